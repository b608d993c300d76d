# dev probe (GPU box): fine-stage tests + fine_match kernel time at 40,960 pairs (split-f16 and plain f16)
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_fine.py -x -q 2>&1 | tail -6
timeout 600 python - <<'PY'
import sys, time, numpy as np, torch
sys.path.insert(0, ".")
from text2loc_amd import synth
from text2loc_amd.engine import Engine
N, Q, K = 11259, 4096, 10
eng = Engine(0)
eng.set_option("profile_events", 1)
eng.fine_load_weights(synth.make_fine_weights(0), class_embed=True, color_embed=True)
cells16 = synth.make_cells(N, seed=17, min_obj=16, max_obj=16)
pk = {k: torch.from_numpy(np.ascontiguousarray(v)).cuda() for k, v in cells16.items() if k != "counts"}
desc = eng.fine_encode_objects(pk)
hints = torch.nn.functional.normalize(torch.randn(Q, 6, 128, device="cuda"), dim=-1)
g = torch.Generator(device="cuda").manual_seed(1)
ci = torch.randint(0, N, (Q * K,), device="cuda", generator=g, dtype=torch.int32)
hi = torch.arange(Q, dtype=torch.int32, device="cuda").repeat_interleave(K)
a = torch.randn(4096, 4096, device="cuda")
for _ in range(30): a @ a
for f16 in (0, 1, 0, 1):
    eng.set_option("encoder_f16", f16)
    for _ in range(3): eng.fine_match(desc, hints, ci, hi)
    torch.cuda.synchronize(); eng.kernel_stats("fine_match")
    for _ in range(10): eng.fine_match(desc, hints, ci, hi)
    torch.cuda.synchronize()
    print("encoder_f16", f16, "fine_match ms", eng.kernel_stats("fine_match"))
PY
