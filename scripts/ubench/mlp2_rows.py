"""ws3d_mlp2_rows (both layers of a head in one kernel) against two library GEMMs; static split of the row tiles vs the ticket counter."""
import torch
from ws3d_amd import compat as C, _lib
lib = _lib.load()
def timeit(f, n=50):
    for _ in range(5): f()
    torch.cuda.synchronize(); a = torch.cuda.Event(True); b = torch.cuda.Event(True); a.record()
    for _ in range(n): f()
    b.record(); torch.cuda.synchronize(); return a.elapsed_time(b) / n * 1e3
for rows, o2 in ((131072, 1), (131072, 40), (16384, 40)):
    x = torch.randn(rows, 128, device='cuda'); w1 = torch.randn(128, 128, device='cuda') / 11; w2 = torch.randn(128, o2, device='cuda') / 11
    b1 = torch.randn(128, device='cuda'); b2 = torch.randn(o2, device='cuda'); out = torch.empty(rows, o2, device='cuda')
    tk = torch.zeros(1, dtype=torch.int32, device='cuda')
    s = torch.cuda.current_stream().cuda_stream
    def static(): lib.ws3d_mlp2_rows(rows, 128, 128, o2, x.data_ptr(), w1.data_ptr(), b1.data_ptr(), 1, w2.data_ptr(), b2.data_ptr(), 0, out.data_ptr(), None, s)
    def ticket():
        tk.zero_()
        lib.ws3d_mlp2_rows(rows, 128, 128, o2, x.data_ptr(), w1.data_ptr(), b1.data_ptr(), 1, w2.data_ptr(), b2.data_ptr(), 0, out.data_ptr(), tk.data_ptr(), s)
    def two(): return torch.addmm(b2, torch._addmm_activation(b1, x, w1, use_gelu=False), w2)
    print(f"rows {rows} o2 {o2}: static {timeit(static):.1f} us, ticket (+ fill) {timeit(ticket):.1f} us, two GEMMs {timeit(two):.1f} us")
