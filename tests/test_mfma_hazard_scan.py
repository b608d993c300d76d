"""Static check of the shipped kernels' ISA for the MFMA -> read hazard that ROCm 7.2 left open across a loop exit (DESIGN.md
§3.6b, "toolchain finding 2"): tools/mfma_hazard_scan.py follows every control-flow path behind every MFMA and counts wait states up
to the first read of the MFMA's destination. The scanner is first checked on two hand-written listings (the shape of the real bug,
and the same listing with the padding the fix inserts); then every translation unit of the library is compiled to assembly (hipcc
cross-compiles; no GPU needed) and must come back with no candidate."""
import glob
import importlib.util
import os
import shutil
import subprocess
from concurrent.futures import ThreadPoolExecutor

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "text2loc_amd", "csrc")


def _scanner():
    spec = importlib.util.spec_from_file_location("mfma_hazard_scan", os.path.join(ROOT, "tools", "mfma_hazard_scan.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


LOOP_EXIT = """
kern:
.LBB0_1:
\tv_mfma_f32_32x32x2_f32 v[2:17], v20, v21, v[2:17]
\ts_add_i32 s4, s4, 1
\ts_cmp_lt_i32 s4, s5
\ts_cbranch_scc1 .LBB0_1
{pad}\tv_mov_b32_e32 v59, v17
\ts_endpgm
"""


def test_scanner_flags_the_loop_exit_copy(tmp_path):
    scan = _scanner().scan
    bad = tmp_path / "bad.s"
    bad.write_text(LOOP_EXIT.format(pad=""))
    hits = scan(str(bad))
    assert len(hits) == 1 and "v_mov_b32_e32 v59, v17" in hits[0][2]
    assert hits[0][5] == 18 - 3  # three instructions behind a 16-pass MFMA: 15 wait states short
    good = tmp_path / "good.s"
    good.write_text(LOOP_EXIT.format(pad="\ts_nop 15\n"))
    assert scan(str(good)) == []
    # back-to-back accumulation (srcC) is exempt; a read of srcA is not
    chain = tmp_path / "chain.s"
    chain.write_text("kern:\n\tv_mfma_f32_32x32x16_bf16 a[0:15], v[0:3], v[4:7], a[0:15]\n\tv_mfma_f32_32x32x16_bf16 a[0:15], v[0:3], v[4:7], a[0:15]\n\ts_endpgm\n")
    assert scan(str(chain)) == []
    chain.write_text("kern:\n\tv_mfma_f32_32x32x16_bf16 v[0:15], v[20:23], v[24:27], v[0:15]\n\tv_mfma_f32_32x32x16_bf16 v[30:45], v[0:3], v[4:7], v[30:45]\n\ts_endpgm\n")
    assert len(scan(str(chain))) == 1


@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"), reason="needs hipcc")
def test_every_translation_unit_is_free_of_candidates(tmp_path):
    scan = _scanner().scan
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    flags = subprocess.run(["make", "-s", "-C", CSRC, "print-flags"], capture_output=True, text=True, check=True).stdout.split()
    units = sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    assert len(units) >= 10

    def dump(src):
        out = str(tmp_path / (os.path.basename(src)[:-4] + ".s"))
        subprocess.run([hipcc, *flags, "-S", "--cuda-device-only", src, "-o", out], check=True, capture_output=True, cwd=CSRC)
        return out

    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as pool:
        listings = list(pool.map(dump, units))
    mfma_seen = 0
    for path in listings:
        mfma_seen += open(path).read().count("v_mfma")
        hits = scan(path)
        assert not hits, f"{os.path.basename(path)}: {len(hits)} candidate(s), first: {hits[0]}"
    assert mfma_seen > 1000  # the scan looked at the real kernels, not at empty listings
