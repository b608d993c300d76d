"""dev: the cached query path List[str] -> ids (bench.py text_head_measure's last part alone)."""
import json
import sys
sys.path.insert(0, ".")
import numpy as np, torch
import bench
from text2loc_amd.engine import Engine
eng = Engine(0)
db = torch.nn.functional.normalize(torch.randn(11259, 256, device="cuda"))
eng.db_set(db)
r = bench.text_head_measure(eng, 11259, 4096)
print(json.dumps(r["cold_query_path"] if "cold_query_path" in r else {k: v for k, v in r.items() if "cache" in k or "cold" in k}, indent=1)[:1500])
