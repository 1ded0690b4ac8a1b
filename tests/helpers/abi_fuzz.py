"""Random descriptors against every compute entry point of the C ABI (run in a subprocess by tests/test_abi.py: a
wild host-side dereference would take the interpreter down).  Without a GPU a descriptor either fails validation
(PV_ERR_INVALID / PV_ERR_UNSUPPORTED) or reaches the launch and comes back as PV_ERR_HIP; nothing may crash, hang or
return a positive status."""
import ctypes as C, random, sys
from pytorchvideo_amd import _lib as L
lib = L.lib()
random.seed(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
ENTRY = {"pv_conv3d": L.Conv3dDesc, "pv_dwconv3d": L.DwConv3dDesc, "pv_se_gate": L.SeGateDesc, "pv_pool3d": L.Pool3dDesc,
         "pv_ingest_ncdhw": L.LayoutDesc, "pv_egress_ncdhw": L.LayoutDesc, "pv_layernorm": L.RowsDesc, "pv_softmax_rows": L.RowsDesc,
         "pv_mean_rows": L.RowsDesc, "pv_add_posenc": L.PosencDesc, "pv_attention": L.AttentionDesc, "pv_add_act": L.AddDesc,
         "pv_token_pool": L.TokenPoolDesc, "pv_roi_align": L.RoiAlignDesc, "pv_ensemble_scores": L.EnsembleDesc}
SMALL = [0, 1, 2, 3, 4, 7, 8, 16, 24, 32, 54, 64, 96, 128, 432, 2048, -1, 1 << 20, (1 << 31) - 1]
def rnd(ft):
    if hasattr(ft, "_length_"):
        return ft(*[rnd(ft._type_) for _ in range(ft._length_)])
    if ft is C.c_void_p: return random.choice([None, 4096, 1 << 33])
    if ft is C.c_float: return random.choice([0.0, 1.0, 0.0625, -1.0, 1e30])
    return random.choice(SMALL)
n, hist = 0, {}
for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 400):
    for name, cls in ENTRY.items():
        d = cls()
        for fn, ft in cls._fields_:
            setattr(d, fn, rnd(ft))
        rc = getattr(lib, name)(C.byref(d), None)
        assert isinstance(rc, int) and rc <= 0, (name, rc)
        n += 1
        hist[rc] = hist.get(rc, 0) + 1
print("ok", n, "calls", sorted(hist.items()))

# ---- second phase: geometrically consistent conv / depthwise descriptors, so that validation passes and the host-side
# routing (kernel choice, tile heuristics, support queries) runs for thousands of shapes; without a GPU the launch
# itself comes back as PV_ERR_HIP
def out(i,k,s,p,d=1): return (i + 2*p - (d*(k-1)+1))//s + 1
hist = {}
P=1<<33
for it in range(6 * (int(sys.argv[2]) if len(sys.argv) > 2 else 400)):
    d=L.Conv3dDesc()
    B=random.choice([1,2,5]); Ti=random.choice([1,2,4,9]); Hi=random.choice([1,4,7,14,33]); Wi=random.choice([1,4,7,14,33])
    cin=random.choice([4,8,16,24,64,96,128,192,432,2048]); cout=random.choice([3,8,24,54,64,80,128,400,2048])
    kt,kh,kw=[random.choice([1,1,3,5,7]) for _ in range(3)]
    st,sh,sw=[random.choice([1,1,2,4]) for _ in range(3)]
    dt,dh,dw=[random.choice([1,1,1,2]) for _ in range(3)]
    pt,ph,pw=[random.choice([0,k//2,1]) for k in (kt,kh,kw)]
    To,Ho,Wo=out(Ti,kt,st,pt,dt),out(Hi,kh,sh,ph,dh),out(Wi,kw,sw,pw,dw)
    if min(To,Ho,Wo)<=0: continue
    c4 = cin==4
    ldx = 4 if c4 else cin + random.choice([0,8])
    ldy = (cout+7)//8*8 + random.choice([0,8,64])
    d.x=d.w=d.y=P; d.scale=random.choice([None,P]); d.shift=random.choice([None,P]); d.residual=random.choice([None,P]); d.a_gate=random.choice([None,None,P])
    d.x_bs=Ti*Hi*Wi*ldx; d.y_bs=To*Ho*Wo*ldy; d.r_bs=d.y_bs; d.ldx=ldx; d.ldy=ldy; d.ldr=ldy
    d.B,d.Ti,d.Hi,d.Wi,d.cin,d.To,d.Ho,d.Wo,d.cout=B,Ti,Hi,Wi,cin,To,Ho,Wo,cout
    d.kt,d.kh,d.kw,d.st,d.sh,d.sw,d.pt,d.ph,d.pw=kt,kh,kw,st,sh,sw,pt,ph,pw
    d.dil_t,d.dil_h,d.dil_w=dt,dh,dw
    d.act=random.choice([0,1,2,3,4]); d.a_act=random.choice([0,0,2]); d.dtype=random.choice([0,1,1]); d.y_f32=random.choice([0,0,1]); d.r_f32=random.choice([0,1])
    if random.random()<0.1: d.dwt_w, d.dwt_k = P, random.choice([3,5])
    if random.random()<0.1: d.c4_wpair = 2
    if random.random()<0.1: d.pos_spatial = P
    if random.random()<0.15:
        d.x2=P; d.x2_scale=P; d.x2_cin=random.choice([8,24,64]); d.x2_ld=d.x2_cin; d.x2_Hi=Ho*random.choice([1,2]); d.x2_Wi=Wo*random.choice([1,2]); d.x2_st=1; d.x2_sh=d.x2_Hi//Ho; d.x2_sw=d.x2_Wi//Wo; d.x2_bs=To*d.x2_Hi*d.x2_Wi*d.x2_ld
    rc=lib.pv_conv3d(C.byref(d),None); assert rc<=0; hist[rc]=hist.get(rc,0)+1
    for q in ("pv_conv3d_dwt_supported","pv_conv3d_x2_supported"):
        getattr(lib,q)(C.byref(d))
    # depthwise
    e=L.DwConv3dDesc(); Cc=random.choice([8,24,54,96,108,432]); ld=(Cc+7)//8*8
    e.x=e.w=e.y=P; e.scale=random.choice([None,P]); e.shift=random.choice([None,P]); e.psum=random.choice([None,P])
    e.x_bs=Ti*Hi*Wi*ld; e.y_bs=To*Ho*Wo*ld; e.ldx=e.ldy=ld; e.B,e.Ti,e.Hi,e.Wi,e.C,e.To,e.Ho,e.Wo=B,Ti,Hi,Wi,Cc,To,Ho,Wo
    e.kt,e.kh,e.kw,e.st,e.sh,e.sw,e.pt,e.ph,e.pw=kt,kh,kw,st,sh,sw,pt,ph,pw
    if (dt,dh,dw)!=(1,1,1): continue
    e.w_mod=random.choice([0,0,Cc//2 if Cc%16==0 else 0]); e.act=random.choice([0,1,2]); e.dtype=random.choice([0,1,1]); e.n_prefix=random.choice([0,0,1])
    if random.random()<0.2: e.pw_w=e.pw_scale=e.pw_shift=P; e.pw_cin=random.choice([8,24,48,64,96]); e.pw_act=1; e.ldx=(e.pw_cin+7)//8*8; e.x_bs=Ti*Hi*Wi*e.ldx
    rc=lib.pv_dwconv3d(C.byref(e),None); assert rc<=0; hist[("dw",rc)]=hist.get(("dw",rc),0)+1
    lib.pv_dwconv3d_psum_blocks(C.byref(e)); lib.pv_dwconv3d_pw_supported(C.byref(e))
print("structured ok", sorted(hist.items(), key=str))

