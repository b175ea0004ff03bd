"""Outputs of single convolutions of every kernel class (rsr_conv3x3 / rsr_conv3x3_res), of the network on one tile and of one
end-to-end call, saved as .npz -- to compare two BUILDS of the library bit for bit:
    RSR_LIB=<build A> python tools/conv_dump.py /tmp/a.npz;  RSR_LIB=<build B> python tools/conv_dump.py /tmp/b.npz
(how round 4 found that the round-3 binary rounded acc*s1 once in some lanes -- a fused v_fma_mix -- and twice in others)."""
import os, sys, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import realsr_ncnn_vulkan_amd as R
from realsr_ncnn_vulkan_amd import synth
out = sys.argv[1]
rng = np.random.default_rng(5)
d = synth.make_model_dir(os.environ.get("RSR_MODELS", "/tmp/rsr_models"), "models-DF2K", 42)
sr = R.RealSR(0); sr.load(os.path.join(d, "x4.param"), os.path.join(d, "x4.bin"))
res = {}
for (cin, cout, h, w, ups, lrelu) in [(64,32,40,70,0,1),(96,32,40,70,0,1),(160,32,40,70,0,1),(64,64,40,70,0,1),(64,64,20,30,1,1),(3,64,40,70,0,0),(64,3,40,70,0,0)]:
    x = rng.standard_normal((cin,h,w)).astype(np.float16); wt=(rng.standard_normal((cout,cin,3,3))/np.sqrt(cin*9)).astype(np.float16).astype(np.float32); b=rng.standard_normal(cout).astype(np.float32)
    res["c%d_%d_%d" % (cin,cout,ups)] = sr.conv3x3(x, wt, b, lrelu=bool(lrelu), upsample2x=bool(ups))
for (cin,cout,own,useres,s1,s2) in [(192,64,1,0,0.2,1.0),(192,64,1,1,0.2,0.2),(64,64,0,1,1.0,1.0)]:
    h,w=40,70
    x = rng.standard_normal((cin,h,w)).astype(np.float16); wt=(rng.standard_normal((cout,cin,3,3))/np.sqrt(cin*9)).astype(np.float16).astype(np.float32); b=rng.standard_normal(cout).astype(np.float32)
    r = rng.standard_normal((cout,h,w)).astype(np.float16) if useres else None
    res["r%d_%d_%d_%d" % (cin,cout,own,useres)] = sr.conv3x3_res(x, wt, b, s1, own_input_residual=bool(own), res=r, s2=s2)
img = synth.make_image(5, 60, 50)
xx = (img.astype(np.float32).transpose(2,0,1)*np.float32(1/255.)).astype(np.float16)
res["net"] = sr.net_forward(xx)
sr.tilesize = 32
res["e2e"] = sr.process(img)
np.savez(out, **res)
