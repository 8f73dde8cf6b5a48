import ctypes, os, torch
here=os.path.dirname(os.path.abspath(__file__))
lib=ctypes.CDLL(os.path.join(here,"libprobe.so"))
print("torch", torch.__version__, "dev", torch.cuda.get_device_name(0))
s=torch.cuda.current_stream().cuda_stream
x=torch.zeros(1000,device="cuda")
rc=lib.probe_fill(ctypes.c_void_p(x.data_ptr()),1000,ctypes.c_float(2.0),ctypes.c_void_p(s)); torch.cuda.synchronize()
print("fill rc",rc, x[:4].tolist(), bool((x==torch.arange(1000,device='cuda')+2).all()))
# side stream
st=torch.cuda.Stream()
with torch.cuda.stream(st):
    y=torch.zeros(1000,device="cuda")
    rc=lib.probe_fill(ctypes.c_void_p(y.data_ptr()),1000,ctypes.c_float(5.0),ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
st.synchronize(); print("side stream ok", bool((y==torch.arange(1000,device='cuda')+5).all()))
g=torch.Generator().manual_seed(0)
A=torch.randint(-4,5,(16,32),generator=g).float(); B=torch.randint(-4,5,(32,16),generator=g).float()
Ab=A.bfloat16().view(torch.int16).cuda(); Bb=B.bfloat16().view(torch.int16).cuda(); C=torch.zeros(16,16,device="cuda")
lib.probe_mfma16(ctypes.c_void_p(Ab.data_ptr()),ctypes.c_void_p(Bb.data_ptr()),ctypes.c_void_p(C.data_ptr()),ctypes.c_void_p(s)); torch.cuda.synchronize()
print("mfma16 ok", bool((C.cpu()==A@B).all()))
A=torch.randint(-4,5,(32,16),generator=g).float(); B=torch.randint(-4,5,(16,32),generator=g).float()
Ab=A.bfloat16().view(torch.int16).cuda(); Bb=B.bfloat16().view(torch.int16).cuda(); C=torch.zeros(32,32,device="cuda")
lib.probe_mfma32(ctypes.c_void_p(Ab.data_ptr()),ctypes.c_void_p(Bb.data_ptr()),ctypes.c_void_p(C.data_ptr()),ctypes.c_void_p(s)); torch.cuda.synchronize()
print("mfma32 ok", bool((C.cpu()==A@B).all()))
o=torch.zeros(64*4,dtype=torch.int16,device="cuda")
lib.probe_tr(ctypes.c_void_p(o.data_ptr()),ctypes.c_void_p(s)); torch.cuda.synchronize()
o=o.cpu().view(64,4)
exp=torch.zeros(64,4,dtype=torch.int16)
for l in range(64):
    g_,i=l>>4,l&15
    for j in range(4): exp[l,j]=(4*g_+j)*64+i
print("tr ok", bool((o==exp).all())); 
if not (o==exp).all(): print(o[:20])
# graph capture of a ctypes launch
gr=torch.cuda.CUDAGraph(); z=torch.zeros(1000,device="cuda")
cs=torch.cuda.Stream()
with torch.cuda.graph(gr, stream=cs):
    lib.probe_fill(ctypes.c_void_p(z.data_ptr()),1000,ctypes.c_float(7.0),ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
z.zero_(); gr.replay(); torch.cuda.synchronize(); print("graph ok", bool((z==torch.arange(1000,device='cuda')+7).all()))
import subprocess; print(subprocess.run("ldd %s | grep -i hip; grep -i hip /proc/%d/maps | awk '{print $6}' | sort -u"%(os.path.join(here,'libprobe.so'),os.getpid()),shell=True,capture_output=True,text=True).stdout)
