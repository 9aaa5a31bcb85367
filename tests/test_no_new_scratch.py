"""No kernel of the built library may acquire a private segment (register spills / stack objects) unnoticed: a spilling kernel
runs its reloads through memory and pays for scratch set-up at every dispatch (round 6: a 27-register spill doubled a warp
variant's launch period; an over-eager residual prefetch put conv1x1_wide_f32_kernel -- 14 % of the fp32 recogniser -- into
scratch until this test's scan showed it).  CPU only: reads the code objects' metadata with the ROCm LLVM tools."""
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SCAN = os.path.join(ROOT, "scripts", "debug", "scratch_scan.py")

# kernels known to have a private segment, none of them launched per batch on a measured path (profiles/r06_warp_lab.txt):
ALLOWED = (
    r"tps_warp_geo_kernel<20, \d, \d, 2, true>",                # run-time geometry WITH grid / index outputs (tests, training forward)
    r"dec_step_persist_kernel<float, \d, (true|false), true>",  # exact-fp32 / bf16x3 persistent decoder, the instantiation for > 64 tokens
    r"dec_step_persist_kernel<unsigned short, \d, false, true>",  # the same for the bf16 head: a 20-byte stack object, no spills
    r"warp_bwd_sample_lds2_kernel<4, 1024, (true|false)>",      # backward for more than 1024 output pixels
    r"warp_bwd_params_kernel<64, 1, false, 1, false>",          # 20-byte stack object, no spills
    r"tps_warp_stream_kernel<32, false, true, 2, (true|false)>",  # plane-streaming warp without a score: 2 registers
)


def test_no_kernel_outside_the_known_list_uses_scratch():
    if not os.path.exists("/opt/rocm/lib/llvm/bin/llvm-readelf"):
        pytest.skip("ROCm LLVM tools not found")
    from tps_pp_amd import build
    build.build()
    out = subprocess.run([sys.executable, SCAN], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr
    head = out.stdout.splitlines()[0]
    m = re.match(r"(\d+) kernels, (\d+) with a private segment", head)
    assert m and int(m.group(1)) > 300, head                     # the scan saw the library's kernels
    offenders = []
    for line in out.stdout.splitlines()[1:]:
        if "B scratch" not in line:
            continue
        if not any(re.search(p, line) for p in ALLOWED):
            offenders.append(line.strip())
    assert not offenders, "kernels with a private segment outside the known list:\n" + "\n".join(offenders)
