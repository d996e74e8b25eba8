"""The steady-state K loop of the 256 x 256 GEMM kernel, read out of hipcc's listing (CPU: hipcc cross-compiles gfx950 without a GPU).

Guards a regression no parity test can see: whether hipcc keeps the two buffer descriptors of the LDS-DMA loads in SGPRs depends on unrelated
code (round 5: an epilogue change made it build them on the VALU -- a v_readfirstlane waterfall loop around every load of the steady state,
every product 3-9 % slower, all tests green).  A healthy loop is 64 MFMAs per K-tile pair, no waterfall loop, no v_readlane (spilled SGPRs),
no VALU beyond one compare of the wave-group flag."""
import importlib.util
import shutil
import subprocess
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
HIPCC = shutil.which("hipcc") or ("/opt/rocm/bin/hipcc" if Path("/opt/rocm/bin/hipcc").exists() else None)


@pytest.mark.skipif(HIPCC is None, reason="hipcc not installed")
def test_steady_k_loop_has_no_waterfall_no_spill_no_valu(tmp_path):
    spec = importlib.util.spec_from_file_location("kloop_stat", ROOT / "tools" / "kloop_stat.py")
    ks = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ks)
    from sprc_amd import build as B
    out = tmp_path / "kloop_probe.s"
    flags = [f for f in B.FLAGS if f not in ("-fPIC",)]
    subprocess.run([HIPCC, *flags, "-S", "--cuda-device-only", "-o", str(out), str(ROOT / "tools" / "kloop_probe.hip")],
                   check=True, capture_output=True, timeout=900)
    lines = out.read_text().split("\n")
    for what, name in ks.KERNELS.items():
        st = ks.steady_loop(lines, name)
        assert st is not None, f"{what}: kernel not in the listing"
        assert st["mfma"] == 64 and st["waterfall"] == 0 and st["readfirstlane"] == 0 and st["readlane"] == 0 and st["valu"] <= 2, (what, st)   # (one v_cmp of the group flag)
        assert st["len"] <= 230, (what, st)          # 216 instructions this round: 64 MFMA, 48 ds_read, 16 loads, 8 barriers, waits and scalar bookkeeping
