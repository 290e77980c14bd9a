import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "gst-plugins-rs_amd"))
import numpy as np, mi355fx
from mi355fx import synth
W,H,B=3840,2160,8
ctx = mi355fx.Context(0)
fr = np.stack([synth.smooth_frame(W,H)]*B)
d = ctx.alloc(fr.nbytes); ctx.h2d(d, fr)
st = synth.HSV_SETTINGS["hue90"]
for bpc in (4, 8, 16, 32, 64, 128, 256, 1024):
    ctx.set_flag(2, bpc)
    ctx.time_hsvfilter_device(d, B, W*H*4, W, H, W*4, "RGBA", st, 5)
    ms = ctx.time_hsvfilter_device(d, B, W*H*4, W, H, W*4, "RGBA", st, 30)
    print("blocks/CU %5d  %.4f ms  %.0f GB/s" % (bpc, ms, 2*fr.nbytes/ms/1e6))
