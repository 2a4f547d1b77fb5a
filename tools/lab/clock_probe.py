import ctypes as C, torch, sys
sys.path.insert(0,'.')
from easygaussiansplatting_amd import _lib
lib=_lib.load()
out=torch.zeros(8,dtype=torch.int64,device='cuda')
st=C.c_void_p(torch.cuda.current_stream().cuda_stream)
for it in (2000, 20000, 20000, 20000):
    e0=torch.cuda.Event(enable_timing=True);e1=torch.cuda.Event(enable_timing=True)
    e0.record(); _lib.check(lib.egs_clock_probe(C.c_void_p(out.data_ptr()), it, st)); e1.record(); torch.cuda.synchronize()
    o=out.cpu().numpy()
    print(it, 'ms', e0.elapsed_time(e1), 'dcyc', o[1]-o[0], 'dreal', o[3]-o[2], 'MHz', (o[1]-o[0])/(o[3]-o[2])*100)
