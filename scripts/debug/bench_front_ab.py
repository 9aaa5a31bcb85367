"""A/B in one process on one box: front_bf16 of the library against an older build of the same entry point
(scripts/ubench/libfront_old.so: `git show <rev>:tps_pp_amd/csrc/tpspp_front_bf16.hip` compiled on its own), interleaved."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tps_pp_amd import TPS_PP, ops, _lib
dev = torch.device("cuda:0"); N = 512
old = ctypes.CDLL(os.path.join(os.path.dirname(__file__), "..", "ubench", "libfront_old.so"))
vp, ci = ctypes.c_void_p, ctypes.c_int
old.tpspp_front_bf16_fwd.argtypes = [vp] * 15 + [ci] * 5 + [vp]; old.tpspp_front_bf16_fwd.restype = ci
m = TPS_PP().eval().to(dev)
x = torch.rand(N, 64, 16, 64, device=dev).bfloat16(); o0 = torch.rand(N, 32, 32, 128, device=dev).bfloat16(); o1 = torch.rand(N, 32, 32, 128, device=dev).bfloat16()
fw = ops.FrontWeightsBf16(m)
outs = [torch.empty(N, 64, 32, 128, device=dev, dtype=torch.bfloat16) for _ in range(3)] + [torch.empty(N, 64, 16, 64, device=dev, dtype=torch.bfloat16)]
def call(L):
    p = lambda t: t.data_ptr()
    rc = L.tpspp_front_bf16_fwd(p(o0), p(o1), p(x), p(fw.w0), p(fw.b0), p(fw.w1), p(fw.b1), p(fw.w2), p(fw.b2), p(fw.wg), p(fw.bg),
                                p(outs[0]), p(outs[1]), p(outs[3]), p(outs[2]), 0, N, 32, 128, 0, torch.cuda.current_stream().cuda_stream)
    assert rc == 0
def t(L, it=20):
    for _ in range(2): call(L)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it): call(L)
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / it * 1e3
new = _lib.lib()
call(old); ref = [o.clone() for o in outs]
for o in outs: o.zero_()
call(new); torch.cuda.synchronize()
print("identical outputs:", all(torch.equal(a, b) for a, b in zip(ref, outs)))
for r in range(2):
    print(f"round {r}: old {t(old):.0f} us | new {t(new):.0f} us")
