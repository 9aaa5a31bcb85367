"""Which kernels of libtpspp_hip.so use scratch (register spills or stack objects)?  A dispatch of a kernel with a private segment
costs ~10 us more on MI355X (measured in round 6 on a variant of the image-pair kernel: 20.3 -> 10.6 us per launch once its 108
bytes of spills were gone), so no kernel that is launched per batch in a hot path may have one.

    python scripts/debug/scratch_scan.py          (CPU only: reads the code objects' metadata)
"""
import glob, os, re, shutil, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
LLVM = "/opt/rocm/lib/llvm/bin"
tmp = tempfile.mkdtemp(prefix="tpspp_co_")
lib = os.path.join(tmp, "libtpspp_hip.so")
shutil.copy(os.path.join(ROOT, "tps_pp_amd", "libtpspp_hip.so"), lib)      # (llvm-objdump extracts next to its input)
subprocess.run([f"{LLVM}/llvm-objdump", "--offloading", lib], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, cwd=tmp)
rows = []
for co in sorted(glob.glob(lib + ".*gfx950")):
    notes = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", co], capture_output=True, text=True).stdout
    for blk in re.split(r"\n\s+- \.agpr_count", notes)[1:]:
        sym = re.search(r"\.symbol:\s+(\S+)\.kd", blk)
        priv = re.search(r"\.private_segment_fixed_size:\s+(\d+)", blk)
        vg = re.search(r"\.vgpr_count:\s+(\d+)", blk)
        sp = re.search(r"\.vgpr_spill_count:\s+(\d+)", blk)
        if sym and priv:
            rows.append((int(priv.group(1)), int(vg.group(1)), int(sp.group(1)) if sp else 0, sym.group(1)))
shutil.rmtree(tmp, ignore_errors=True)
print(f"{len(rows)} kernels, {sum(1 for r in rows if r[0])} with a private segment")
for priv, vg, sp, sym in sorted(rows, reverse=True):
    if priv:
        name = subprocess.run(["c++filt", sym], capture_output=True, text=True).stdout.strip()
        print(f"{priv:5d} B scratch  {vg:3d} VGPRs  {sp:3d} spilled  {name[:160]}")
