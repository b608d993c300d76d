"""dev probe: per-call time and counters of the search on clustered databases (where does the cliff start, and what pays for it)."""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from text2loc_amd import synth
from text2loc_amd.engine import Engine

N, Q, K, D = 11259, 4096, 10, 256
rs = np.random.default_rng(31)
C = 64
cent = synth.unit_rows(rs.standard_normal((C, D)))
noise = synth.unit_rows(rs.standard_normal((N, D)))
member = rs.integers(0, C, size=N)
alphas = [float(a) for a in sys.argv[1:]] or [2.0, 3.0, 5.0, 10.0]
for alpha in alphas:
    cc = cent[member]
    dbc = synth.unit_rows(alpha * cc + noise).astype(np.float32)
    tgt = rs.integers(0, N, size=Q)
    spread = float(np.linalg.norm(dbc - cc * (dbc * cc).sum(1, keepdims=True), axis=1).mean())
    q = synth.unit_rows(dbc[tgt].astype(np.float64) + 0.25 * spread * synth.unit_rows(rs.standard_normal((Q, D)))).astype(np.float32)
    e = Engine(0)
    e.db_set(torch.from_numpy(dbc).cuda())
    dq = torch.from_numpy(q).cuda()
    print(f"alpha {alpha}")
    for auto, wide in ((1, 1024), (1, 0)):
        e.set_option("search_auto", 0)
        e.set_option("search_auto", auto)
        e.set_option("search_wide_repair", wide)
        for i in range(12):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            e.search(dq, K)
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) * 1e6
            c = e.search_counters()
            print(f"  auto {auto} wide {wide} call {i}: {dt:8.1f} us  rescored {c['rescored']} to_fb {c['to_fallback_kernel']} valu_exact {c['valu_exact_scans']} probe {c['probe']} deferred {c['deferred_to_mfma_exact']} served {c['mfma_exact_served']}")
        e.set_option("profile_events", 1)
        for nme in ("search_scan", "search_rerank", "search_exact"):
            e.kernel_stats(nme)
        for i in range(6):
            e.search(dq, K)
        torch.cuda.synchronize()
        print("   kernels ms:", {nme: e.kernel_stats(nme) for nme in ("search_scan", "search_rerank", "search_exact")})
        e.set_option("profile_events", 0)
    e.close()
