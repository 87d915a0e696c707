import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from multi_robot_slam_separators_amd import lib, synth
from oracle import netvlad_torch as ref
w = ref.random_weights(3, clusters=64, pca_dim=512)
rng = np.random.default_rng(5)
image = rng.uniform(0, 255, size=(240, 320, 3)).astype(np.float32)
want = ref.netvlad(image, w)
f = lib.SeparatorFinder(synth.camera_params(), device=0)
f.set_stream(torch.cuda.current_stream().cuda_stream)
f.netvlad_load(w)
d_img = torch.from_numpy(image).cuda(); d_out = torch.zeros(512, device="cuda")
f.netvlad_infer_device(d_img.data_ptr(), 320, 240, d_out.data_ptr(), 512); torch.cuda.synchronize()
got = d_out.cpu().numpy()
print(os.environ.get("SF_CNN_FP32", "split-f16"), "max abs err vs torch fp32 CPU: %.3e  (descriptor entries ~ %.3e)" % (np.abs(got - want).max(), np.abs(want).mean()))
f.close()
