"""Dev tool (GPU box): longer randomised sweep of t2l_search against the C oracle — random N / Q / K / row offsets / data kinds,
stream-ordered and pipelined (search_lanes) — than tests/test_gpu_search.py::test_randomized_shapes_and_data_kinds runs.
python tools/fuzz_search.py [n_draws] [seed]"""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import c_oracle
from text2loc_amd import synth
from text2loc_amd.engine import Engine

def main(n_draws=150, seed=7):
    rng = np.random.default_rng(seed)
    eng = Engine(0)
    bad = 0
    for it in range(n_draws):
        n = int(rng.choice([1, 5, 31, 32, 33, 64, 100, 257, 511, 1000, 2049, 4097, 5000, 11259, 20000]))
        q = int(rng.choice([1, 2, 3, 4, 7, 8, 12, 16, 17, 31, 64, 65, 255, 256, 257, 700, 1024, 4096]))
        k = int(rng.choice([1, 3, 5, 10, 11, 26]))
        kind = int(rng.integers(0, 4))
        lanes = int(rng.choice([1, 1, 2, 3, 4]))
        db, qs, _ = synth.make_retrieval_problem(n, q, seed=int(rng.integers(1 << 30)), noise=float(rng.choice([0.1, 0.5, 2.0])))
        if kind == 1:
            base = synth.unit_rows(rng.standard_normal((1, 256)))
            db = synth.unit_rows(base + 10 ** rng.uniform(-4, -2) * rng.standard_normal((n, 256))).astype(np.float32)
        elif kind == 2:
            db = (db * np.float32(10 ** rng.uniform(-8, 8))).astype(np.float32)
            qs = (qs * np.float32(10 ** rng.uniform(-8, 8))).astype(np.float32)
        elif kind == 3 and n > 40:
            db[n // 2:n // 2 + min(20, n // 4)] = db[:min(20, n // 4)]
        off = int(rng.integers(0, 1000))
        eng.set_option("search_lanes", lanes)
        eng.db_set(torch.from_numpy(db).cuda(), off)
        dq = torch.from_numpy(qs).cuda()
        outs = [eng.search(dq, k, join=False) for _ in range(3 if lanes > 1 else 1)]
        eng.search_join()
        torch.cuda.synchronize()
        ridx, rsc = c_oracle.retrieve_topk(db, qs, k)
        kk = ridx.shape[1]
        scale = max(1.0, float(np.abs(rsc).max()))
        for idx, sc in outs:
            idx, sc = idx.cpu().numpy().astype(np.int64), sc.cpu().numpy()
            ok = np.abs(sc[:, :kk] - rsc).max() <= 1e-12 * scale
            for a, b in np.argwhere(idx[:, :kk] != ridx + off):
                ok = ok and abs(sc[a, b] - rsc[a, b]) <= 1e-13 * scale
            if not ok:
                bad += 1
                print("MISMATCH", dict(n=n, q=q, k=k, kind=kind, lanes=lanes, off=off))
    print(f"{n_draws} draws, {bad} mismatches")
    return bad

if __name__ == "__main__":
    sys.exit(1 if main(*(int(a) for a in sys.argv[1:])) else 0)
