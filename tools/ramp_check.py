import sys, time, ctypes, os
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from katago_amd import capi, modelgen, nninterface as nn
from bench import synthetic_rows
lib = capi.load_library(); nn.globalInitialize()
p = "/tmp/ramp_b18.bin"; modelgen.write_model(p, "b18c384nbt", seed=1)
model = nn.loadModelFile(p); ctx = nn.createComputeContext([0], 19, 19, precision="bf16")
h = nn.createComputeHandle(ctx, model, 256, True, 0)
B, S = 256, 361
sp, gl = synthetic_rows(B, 1)
d_sp, d_gl = torch.from_numpy(sp).cuda(), torch.from_numpy(gl).cuda()
sym = (np.arange(B) % 8).astype(np.int32); opt = np.zeros(B, np.float32)
d_pol = torch.empty((B, S + 1), device="cuda"); d_val = torch.empty((B, 3), device="cuda"); d_sc = torch.empty((B, 6), device="cuda"); d_own = torch.empty((B, S), device="cuda")
torch.cuda.synchronize()
def step():
    nn.getOutputDevice(h, d_sp.data_ptr(), d_gl.data_ptr(), sym, opt, d_pol.data_ptr(), d_val.data_ptr(), d_sc.data_ptr(), d_own.data_ptr(), sync=False)
t00 = time.perf_counter()
for chunk in range(16):
    t0 = time.perf_counter()
    for _ in range(10): step()
    h.sync()
    dt = time.perf_counter() - t0
    print("chunk %2d at %.2fs: %.0f evals/s" % (chunk, time.perf_counter() - t00, 10 * B / dt), flush=True)
