"""Census behind kernels_frag.h: share of listed atom pairs that lie inside the cutoff for cluster-pair lists of several shapes on
the all-mobile S23k system (Hilbert order, cutoff 1.0 nm), and for fragments of three consecutive atoms (a water = one fragment).
   python scripts/census_clusters.py"""
import numpy as np, sys, time
sys.path.insert(0,'/root/repo')
from blues_amd import systems
s,_v = systems.s23k(frozen=False)
x = np.asarray(s.positions).reshape(-1,3)
box = np.asarray(s.box).reshape(-1)
print(x.shape, box)
n=len(x)
# hilbert sort: use 10-bit morton as a proxy? implement hilbert via simple library-free algorithm
def hilbert3(X, bits=10):
    # X: (n,3) ints; Skilling's transform
    X = X.copy().astype(np.uint32)
    M = np.uint32(1 << (bits-1))
    Q = M
    while Q > 1:
        P = np.uint32(Q-1)
        for i in range(3):
            m = (X[:,i] & Q) != 0
            X[m,0] ^= P
            t = (X[:,0]^X[:,i]) & P
            t[m] = 0
            X[:,0]^=t; X[:,i]^=t
        Q >>= 1
    for i in range(1,3): X[:,i]^=X[:,i-1]
    t = np.zeros(len(X),np.uint32)
    Q = M
    while Q>1:
        m = (X[:,2]&Q)!=0
        t[m]^=np.uint32(Q-1)
        Q>>=1
    for i in range(3): X[:,i]^=t
    key = np.zeros(len(X),np.uint64)
    for b in range(bits-1,-1,-1):
        for i in range(3):
            key = (key<<np.uint64(1)) | ((X[:,i]>>np.uint32(b))&1).astype(np.uint64)
    return key
fr = (x/box) % 1.0
key = hilbert3(np.minimum(1023,(fr*1024).astype(np.int64)))
order = np.argsort(key, kind='stable')
xs = x[order]
def mi(d): return d - box*np.round(d/box)
def census(CI, CJ, rl, rc=1.0):
    nci = n//CI; ncj = n//CJ
    # cluster bbox relative to first atom
    def bb(C, nc):
        p = xs[:nc*C].reshape(nc,C,3)
        rel = mi(p - p[:,:1,:])
        lo = rel.min(1); hi = rel.max(1)
        c = p[:,0,:] + 0.5*(lo+hi); h = 0.5*(hi-lo)
        return c,h
    ci_c,ci_h = bb(CI,nci); cj_c,cj_h = bb(CJ,ncj)
    tot=0
    for a in range(0,nci,256):
        d = np.abs(mi(ci_c[a:a+256,None,:]-cj_c[None,:,:])) - ci_h[a:a+256,None,:] - cj_h[None,:,:]
        d = np.maximum(d,0); d2=(d*d).sum(-1)
        tot += (d2 < rl*rl).sum()
    return tot*CI*CJ
# in-range atom pairs (full count, both directions)
from scipy.spatial import cKDTree
t = cKDTree(fr*box, boxsize=box)
np_in = t.count_neighbors(t, 1.0) - n
print("in-range ordered pairs", np_in, "per atom", np_in/n)
for rl in (1.0,1.06,1.12,1.2):
    npl = t.count_neighbors(t, rl) - n
    print("rl",rl,"per-atom list eff", np_in/npl)
    for CI,CJ in ((8,8),(4,16),(16,4),(4,8),(8,4),(4,4),(64,8),(64,1),(64,64),(32,32),(16,16)):
        le = census(CI,CJ,rl)
        print("  ",CI,CJ,"lane-evals(full)",le,"eff",np_in/le)

print("---- fragments of 3 consecutive atoms in caller order (water = molecule)")
F = n//3
xf = (fr*box).reshape(F,3,3)
# unwrap each fragment around its first atom
rel = mi(xf - xf[:,:1,:]); cen = xf[:,0,:] + rel.mean(1); rad = np.sqrt(((rel-rel.mean(1,keepdims=True))**2).sum(-1)).max(1)
print("fragment radius: mean %.3f max %.3f"%(rad.mean(), rad.max()))
tf = cKDTree(cen % box, boxsize=box)
for rl in (1.0,1.04,1.08,1.12,1.2,1.3):
    # conservative: centroid distance < rl + rad_i + rad_j  ~ use max radius of water 0.07 -> approximate with uniform rmax for waters
    # exact criterion: any atom pair < rl.  approximate via centroid test with per-pair radii: do it by query_ball with rl+2*rad.max then filter
    pairs = tf.query_pairs(rl + 2*rad.max(), output_type='ndarray')
    d = mi(cen[pairs[:,0]] - cen[pairs[:,1]])
    dist = np.sqrt((d*d).sum(-1))
    keep_cons = dist < rl + rad[pairs[:,0]] + rad[pairs[:,1]]
    # exact any-atom-pair test
    pa = xf[pairs[:,0]]; pb = xf[pairs[:,1]]
    dd = mi(pa[:,:,None,:]-pb[:,None,:,:]); d2 = (dd*dd).sum(-1).reshape(len(pairs),9)
    keep_exact = d2.min(1) < rl*rl
    inr = (d2 < 1.0).sum()
    print("rl %.2f: conservative list %d pairs/frag %.1f eff %.3f | exact list pairs/frag %.1f eff %.3f" % (rl, keep_cons.sum(), 2*keep_cons.sum()/F, inr/(9*keep_cons.sum()), 2*keep_exact.sum()/F, inr/(9*keep_exact.sum())))
