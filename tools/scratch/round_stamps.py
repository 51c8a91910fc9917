import sys, ctypes as C
sys.path.insert(0, "visual-odometry-project_amd")
import numpy as np
from vo import _native, synthetic
ctx = _native.Context(0)
img, depth, T, K = synthetic.render(3, 1241, 1376)
H, W = img.shape
d_img = ctx.to_device(img); d_sc = ctx.alloc(H*W*8); d_kp = ctx.alloc(2000*16)
nblk = ((W+63)//64)*((H+31)//32)
d_st = ctx.alloc(nblk*24*8)
ctx.harris_response_dev(d_img, H, W, 9, 0.09, d_sc)
for it in range(3): ctx.nms_keypoints_dev(d_sc, H, W, 2000, 5, d_kp)
ctx.sync()
lib = _native.load(); lib.vo_debug_set_stamps.argtypes=[C.c_void_p, C.c_void_p]
lib.vo_debug_set_stamps(ctx._h, C.c_void_p(d_st))
ctx.nms_keypoints_dev(d_sc, H, W, 2000, 5, d_kp); ctx.sync()
st = ctx.download(d_st, (nblk, 24), np.uint64).astype(np.int64)
n = st[:,23]
print("candidates per tile: mean %.1f max %d, tiles with n>0: %d" % (n.mean(), n.max(), (n>0).sum()))
act = st[n>0]
t0 = act[:,0].min()
print("kernel span (cycles):", act[:,1:18].max() - t0)
iters = (act[:,2:18] > 0).sum(axis=1)
print("iterations per tile: mean %.2f max %d" % (iters.mean(), iters.max()))
print("load phase mean cycles", (act[:,1]-act[:,0]).mean())
last = np.array([row[2:18][row[2:18]>0].max() for row in act])
dur = last - act[:,0]
o = np.argsort(-dur)[:8]
for k in o: print("tile dur %d cycles n=%d iters=%d start+%d per-iter:" % (dur[k], act[k,23], iters[k], act[k,0]-t0), np.diff(np.concatenate([[act[k,1]], act[k,2:2+iters[k]]])))
print("median tile dur", np.median(dur), "sum/256CUs", dur.sum()/256)
