import sys, json
sys.path.insert(0, "/root/repo")
import torch
import bench
from text2loc_amd.engine import Engine
from text2loc_amd import synth
import numpy as np
eng = Engine(0)
db, qs, _ = synth.make_retrieval_problem(11259, 64, seed=1)
eng.db_set(torch.from_numpy(db).cuda())
r = bench.text_head_measure(eng, 11259, 4096)
print(json.dumps({k: r[k] for k in ("d1024_layer_plus_linear_ms", "d256_half", "total_ms", "cold_query_path")}, indent=1))
