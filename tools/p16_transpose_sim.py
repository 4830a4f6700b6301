import numpy as np
rng=np.random.default_rng(0)
B=rng.integers(0,256,size=(16,16),dtype=np.uint8)   # B[i][c]: lane i byte c
X=[[int.from_bytes(bytes(B[i,4*k:4*k+4]),'little') for k in range(4)] for i in range(16)]
def perm(a,b,sel):
    src=list(b.to_bytes(4,'little'))+list(a.to_bytes(4,'little'))
    out=[]
    for j in range(4):
        s=(sel>>(8*j))&0xff
        out.append(0 if s==0x0c else src[s])
    return int.from_bytes(bytes(out),'little')
def dpp(src_of_lane, ctrl):
    # returns function lane->source lane or None
    def f(i):
        if ctrl<0x100:
            q=i&~3; return q+((ctrl>>(2*(i&3)))&3)
        if 0x101<=ctrl<=0x10f: s=i+(ctrl-0x100); return s if s<16 else None
        if 0x111<=ctrl<=0x11f: s=i-(ctrl-0x110); return s if s>=0 else None
        if 0x121<=ctrl<=0x12f: return (i-(ctrl-0x120))%16
    return f
def upd(old,src,ctrl,bank):
    f=dpp(None,ctrl); out=[]
    for i in range(16):
        s=f(i)
        if ((bank>>(i>>2))&1) and s is not None: out.append(src[s])
        else: out.append(old[i])
    return out
col=lambda k:[X[i][k] for i in range(16)]
def setcol(k,v):
    for i in range(16): X[i][k]=v[i]
selA=[0x03070105 if i&1 else 0x06020400 for i in range(16)]
selB=[0x03020706 if i&2 else 0x05040100 for i in range(16)]
for k in range(4):
    P=upd([0]*16,col(k),0xB1,0xf); setcol(k,[perm(P[i],X[i][k],selA[i]) for i in range(16)])
for k in range(4):
    P=upd([0]*16,col(k),0x4E,0xf); setcol(k,[perm(P[i],X[i][k],selB[i]) for i in range(16)])
for a,b in ((0,1),(2,3)):
    t=col(a); na=upd(col(a),col(b),0x114,0xA); nb=upd(col(b),t,0x104,0x5); setcol(a,na); setcol(b,nb)
for a,b in ((0,2),(1,3)):
    t=col(a); na=upd(col(a),col(b),0x128,0xC); nb=upd(col(b),t,0x128,0x3); setcol(a,na); setcol(b,nb)
T=np.array([[b for k in range(4) for b in X[i][k].to_bytes(4,'little')] for i in range(16)],dtype=np.uint8)
print("transpose ok:", np.array_equal(T,B.T))
