import os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
import realsr_ncnn_vulkan_amd as R
from realsr_ncnn_vulkan_amd import synth
d = synth.make_model_dir(os.environ.get("RSR_MODELS", "/tmp/rsr_models"), "models-DF2K", 42)
sr = R.RealSR(0); sr.load(os.path.join(d, "x4.param"), os.path.join(d, "x4.bin")); sr.tilesize = 200
w, h = 7680, 4320
img = synth.make_image(3, w, h)
t = time.time(); out = sr.process(img); dt = time.time() - t
print("8K frame: %.2f s, %.1f Mpix/s out, shape %s" % (dt, out.size / 3e6 / dt, out.shape))
# tile locality: the top-left 3x3 tiles of the big frame vs the same crop processed alone (interior tile (1,1) sees the same halo)
crop = img[:600, :600]
oc = sr.process(crop)
a = out[800:1600, 800:1600]; b = oc[800:1600, 800:1600]
print("interior tile identical to the crop's:", bool((a == b).all()))
t = time.time(); out2 = sr.process(img, out=out); dt = time.time() - t
print("second call: %.2f s" % dt)
sr.close()
