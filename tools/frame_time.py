"""C2 frame time through rsr_process_device, unprofiled (kernels back to back) and profiled (HIP events around every launch):
    RSR_LIB=<build> python tools/frame_time.py [option=value ...]      -> one line"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import realsr_ncnn_vulkan_amd as R  # noqa: E402
from realsr_ncnn_vulkan_amd import synth  # noqa: E402

d = synth.make_model_dir(os.environ.get("RSR_MODELS", "/tmp/rsr_models"), "models-DF2K", 42)
sr = R.RealSR(0)
sr.load(os.path.join(d, "x4.param"), os.path.join(d, "x4.bin"))
sr.tilesize = 200
for kv in sys.argv[1:]:
    k, v = kv.split("=")
    sr.set_option(k, int(v))
w, h = 1920, 1080
img = synth.make_image(3, w, h)
d_in = torch.from_numpy(img).cuda()
d_out = torch.empty((h * 4, w * 4, 3), dtype=torch.uint8, device="cuda")
for _ in range(3):
    sr.process_device(d_in.data_ptr(), w, h, 3, d_out.data_ptr())
torch.cuda.synchronize()
res = []
for rep in range(3):
    t = time.perf_counter()
    n = 6
    for _ in range(n):
        sr.process_device(d_in.data_ptr(), w, h, 3, d_out.data_ptr())
    torch.cuda.synchronize()
    res.append((time.perf_counter() - t) / n * 1e3)
sr.set_profiling(True)
sr.get_profile(reset=True)
for _ in range(3):
    sr.process_device(d_in.data_ptr(), w, h, 3, d_out.data_ptr())
p = sr.get_profile()
sr.set_profiling(False)
print("frame ms (3 x 6 frames): %s | profiled: conv sum %.2f ms/frame | checksum %d" % (
    " ".join("%.2f" % r for r in res), p["conv_ms"] / 3, int(d_out[::97, ::89].to(torch.int64).sum().item())))
sr.close()
