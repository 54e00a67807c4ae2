"""Probe v_mfma_scale_f32_16x16x128_f8f6f4 (fp8 e4m3 operands, unit scales) through mila_cdna4_selftest_mfma_fp8."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from mila_amd import capi
lib = capi.load()
q = lambda x: torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).to(torch.float8_e4m3fn).view(torch.uint8).cuda()
def run(A, B, mode=0):
    Cd = torch.zeros(256, dtype=torch.float32, device="cuda")
    Ad, Bd = q(A), q(B)          # keep the device buffers alive across the call
    capi.check(lib.mila_cdna4_selftest_mfma_fp8(C.c_void_p(Cd.data_ptr()), C.c_void_p(Ad.data_ptr()), C.c_void_p(Bd.data_ptr()), mode,
                                                 C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    return Cd.cpu().numpy().reshape(16, 16)
ones = np.ones((16, 128), dtype=np.float32)
print("ones x ones:", np.unique(run(ones, ones)))
A = np.zeros((16, 128), dtype=np.float32); A[3, :] = 1.0
print("A row 3 = 1, B ones -> nonzero rows/cols of C:", np.nonzero(run(A, ones).sum(1))[0], np.nonzero(run(A, ones).sum(0))[0])
B = np.zeros((16, 128), dtype=np.float32); B[5, :] = 1.0
print("A ones, B row 5 = 1 -> nonzero rows/cols of C:", np.nonzero(run(ones, B).sum(1))[0], np.nonzero(run(ones, B).sum(0))[0])
A = np.zeros((16, 128), dtype=np.float32); A[:, 7] = 2.0
print("A col 7 = 2, B ones:", np.unique(run(A, ones)))
rng = np.random.default_rng(0)
vals = np.array([-3, -2, -1.5, -1, -0.5, 0, 0.5, 1, 1.5, 2, 3, 4], dtype=np.float32)
A = rng.choice(vals, (16, 128)); B = rng.choice(vals, (16, 128))
exp = A.astype(np.float64) @ B.astype(np.float64).T
for mode in (0, 1):
    got = run(A, B, mode)
    print("mode", mode, "max|C - A B^T| =", np.abs(got - exp).max(), " vs transposed:", np.abs(got - exp.T).max())
