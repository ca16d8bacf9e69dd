"""Debug helper: at_op_gemm_split on random data, printing where the output differs from numpy float64 (not part of the product)."""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from audiotoken_amd import _cabi

lib = _cabi.load()
dev = torch.device("cuda:0")
M, N, K = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
kernel = int(sys.argv[4])
rng = np.random.default_rng(0)
x = rng.standard_normal((M, K)).astype(np.float32)
w = (rng.standard_normal((N, K)) * 0.05).astype(np.float32)
ref = x.astype(np.float64) @ w.astype(np.float64).T
xd, wd = torch.from_numpy(x).to(dev), torch.from_numpy(w).to(dev)
out = torch.full((M, N), float("nan"), dtype=torch.float32, device=dev)
nbytes = ((M + 255) // 256 * 256 + N) * K * 2 * 2
ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
st = torch.zeros(1, dtype=torch.int32, device=dev)
rc = lib.at_op_gemm_split(xd.data_ptr(), wd.data_ptr(), 0, out.data_ptr(), M, N, K, 1, float(np.abs(w).max()), kernel, ws.data_ptr(), nbytes, st.data_ptr(),
                          _cabi.current_stream_handle(dev))
torch.cuda.synchronize()
got = out.cpu().numpy().astype(np.float64)
print("rc", rc, "status", int(st.item()), "nan frac", np.isnan(got).mean())
bad = ~(np.abs(got - ref) < 1e-3)
print("bad frac", bad.mean())
for mt in range(0, M, 64):
    row = []
    for nt in range(0, N, 64):
        row.append(f"{bad[mt:mt + 64, nt:nt + 64].mean():.2f}")
    print(mt, " ".join(row))
i = np.argwhere(bad)
if len(i):
    m, n = i[0]
    print("first bad", m, n, got[m, n], ref[m, n], "ratio", got[m, n] / ref[m, n])
    print(got[m, n:n + 8], ref[m, n:n + 8])
