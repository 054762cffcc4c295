import ctypes as C, torch, sys
sys.path.insert(0, '/root/repo')
import abcnet_amd
from abcnet_amd import _lib as L
lib = L.load()
for (B,H,W) in [(16,384,384),(16,192,192),(16,96,96),(16,48,48),(16,24,24),(16,12,12)]:
    st = torch.randn(B,H,W,2,device='cuda'); du = torch.randn(B,H,W,device='cuda')
    w7 = torch.randn(98,device='cuda')*0.1; b7 = torch.randn(1,device='cuda')
    sa = torch.empty(B,H,W,device='cuda'); dst = torch.empty(B,H,W,2,device='cuda')
    dw, db = torch.empty(98,device='cuda'), torch.empty(1,device='cuda')
    d = L.CbamConv7Desc()
    d.st,d.w7,d.b7,d.sa,d.du,d.dst = st.data_ptr(),w7.data_ptr(),b7.data_ptr(),sa.data_ptr(),du.data_ptr(),dst.data_ptr()
    d.B,d.H,d.W = B,H,W
    nb = lib.abc_cbam_conv7_blocks(C.byref(d)); part = torch.empty(nb,99,device='cuda')
    d.dw_partial,d.dw7,d.db7 = part.data_ptr(),dw.data_ptr(),db.data_ptr()
    s = torch.cuda.current_stream().cuda_stream
    for fn,name in ((lib.abc_cbam_conv7_fwd,'fwd'),(lib.abc_cbam_conv7_bwd,'bwd')):
        for _ in range(3): fn(C.byref(d), s)
        e0,e1 = torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): fn(C.byref(d), s)
        e1.record(); torch.cuda.synchronize()
        print(f"{H}x{W} {name}: {e0.elapsed_time(e1)/20*1000:.1f} us", flush=True)
