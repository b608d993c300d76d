"""Dev tool (GPU box): randomised sweep of t2l_merge_gathered / t2l_merge_topk against the host merge — random part counts, k, query
counts, short lists (-1 / -inf tails), whole empty parts, exact score ties across parts (row ids break them), scores spanning many
magnitudes and negative scores (the kernel's float32 lower bound must stay a lower bound).   python tools/fuzz_merge.py [n_draws] [seed]"""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from text2loc_amd.engine import Engine
from text2loc_amd.sharded import merge_topk_host


def main(n_draws=200, seed=3):
    rng = np.random.default_rng(seed)
    eng = Engine(0)
    bad = 0
    for it in range(n_draws):
        P = int(rng.integers(1, 9))
        K = int(rng.choice([1, 3, 5, 10, 16, 26]))
        if P * K > 256:
            continue
        Q = int(rng.choice([1, 3, 64, 257, 1000, 4096]))
        kind = int(rng.integers(0, 5))
        sc = rng.standard_normal((P, Q, K))
        if kind == 1:
            sc = np.round(sc, 1)  # many exact ties
        elif kind == 2:
            sc = sc * 10.0 ** rng.integers(-30, 30, size=(P, Q, 1))
        elif kind == 3:
            sc = 1.0 + 1e-9 * sc  # packed far tighter than float32 resolves
        elif kind == 4:
            sc = -np.abs(sc) * 1e-3
        sc = -np.sort(-sc, axis=2)  # every part's list is sorted (best first)
        ids = np.empty((P, Q, K), dtype=np.int32)
        for p in range(P):
            ids[p] = np.cumsum(rng.integers(1, 100, size=(Q, K)), axis=1) + p * 5000  # unique inside a list and across parts
            # equal scores inside a part: lower row first (what the per-shard search returns)
            order = np.lexsort((ids[p], -sc[p]), axis=1)
            ids[p] = np.take_along_axis(ids[p], order, axis=1)
            sc[p] = np.take_along_axis(sc[p], order, axis=1)
        valid = rng.integers(0, K + 1, size=(P, Q)) if rng.random() < 0.5 else np.full((P, Q), K)
        if rng.random() < 0.3:
            valid[rng.integers(0, P)] = 0  # an empty shard
        tail = np.arange(K)[None, None, :] >= valid[:, :, None]
        ids[tail] = -1
        sc[tail] = -np.inf
        if it % 3 == 1:  # UNSORTED parts, invalid entries anywhere (the public t2l_merge_topk promises no order inside a part)
            perm = rng.permuted(np.tile(np.arange(K), (P, Q, 1)), axis=2)
            ids, sc = np.take_along_axis(ids, perm, axis=2), np.take_along_axis(sc, perm, axis=2)
        ref_i, ref_s = merge_topk_host(ids, sc, K)
        buf, _, _, bb, so = eng.result_block(Q, K, "cuda", parts=P)
        for p in range(P):
            buf[p, :Q * K * 4].view(torch.int32).view(Q, K).copy_(torch.from_numpy(ids[p]))
            buf[p, so:so + Q * K * 8].view(torch.float64).view(Q, K).copy_(torch.from_numpy(sc[p]))
        gi, gs = eng.merge_gathered(buf.view(-1), bb, so, P, Q, K)
        ti, ts = eng.merge_topk(torch.from_numpy(ids).cuda(), torch.from_numpy(sc).cuda())
        torch.cuda.synchronize()
        for name, (a, b) in {"gathered": (gi, gs), "topk": (ti, ts)}.items():
            ok = np.array_equal(a.cpu().numpy().astype(np.int64), ref_i) and np.array_equal(b.cpu().numpy(), ref_s)
            if not ok:
                bad += 1
                print("MISMATCH", name, "draw", it, "P", P, "K", K, "Q", Q, "kind", kind)
    # a duplicated (score, id) pair (row ids are meant to be unique across parts): both copies come out, no slot is left unwritten
    ids = np.array([[[7, 9, 11]], [[7, 20, 21]]], dtype=np.int32)
    sc = np.array([[[0.9, 0.5, 0.1]], [[0.9, 0.6, 0.2]]])
    di, ds = eng.merge_topk(torch.from_numpy(ids).cuda(), torch.from_numpy(sc).cuda())
    if di.cpu().numpy().tolist() != [[7, 7, 20]] or ds.cpu().numpy().tolist() != [[0.9, 0.9, 0.6]]:
        bad += 1
        print("MISMATCH duplicate pair", di.cpu().numpy(), ds.cpu().numpy())
    print("draws", n_draws, "mismatches", bad)
    return bad


if __name__ == "__main__":
    sys.exit(1 if main(*(int(a) for a in sys.argv[1:])) else 0)
