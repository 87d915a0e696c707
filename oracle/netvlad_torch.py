"""CPU REFERENCE (test infrastructure, never imported by the product) of the NetVLAD descriptor network the reference
runs in DataHandler.compute_descriptors (data_handler.py:59-70,143-164): `nets.vgg16NetvladPca` of netvlad_tf_open,
which the reference imports but does not vendor -- PARITY UNPINNED, restated from the published definition
(python/netvlad_tf/nets.py, layers.py) in plain PyTorch float32 on the CPU, the form the tier rules allow for a
floating-point kernel.  Weight layouts are TensorFlow's (what sf_netvlad_load takes)."""
import numpy as np
import torch
import torch.nn.functional as F

from multi_robot_slam_separators_amd import _abi

RELU = [True, False, True, False, True, True, False, True, True, False, True, True, False]
POOL = [False, True, False, True, False, False, True, False, False, True, False, False, False]


def random_weights(seed, clusters=64, pca_dim=4096):
    """He-style random weights of the right shapes (there is no checkpoint to load here)."""
    rng = np.random.default_rng(seed)
    w = {"conv_kernel": [], "conv_bias": []}
    for ci, co in _abi.VGG16_CONVS:
        w["conv_kernel"].append((rng.standard_normal((3, 3, ci, co)) * np.sqrt(2.0 / (9 * ci))).astype(np.float32))
        w["conv_bias"].append((rng.standard_normal(co) * 0.05).astype(np.float32))
    w["average_rgb"] = np.array([123.68, 116.779, 103.939], np.float32)
    w["assignment"] = (rng.standard_normal((512, clusters)) * 2.0).astype(np.float32)
    w["cluster_centers"] = (rng.standard_normal((512, clusters)) * 0.05).astype(np.float32)
    w["wpca_kernel"] = (rng.standard_normal((512 * clusters, pca_dim)) / np.sqrt(512 * clusters)).astype(np.float32)
    w["wpca_bias"] = (rng.standard_normal(pca_dim) * 0.01).astype(np.float32)
    return w


def conv5_3(image_hwc, w, upto=13):
    """VGG16 trunk: image float32 [H, W, 3] (RGB) -> [h, w, C] activations after layer `upto` (13 = conv5_3)."""
    x = torch.from_numpy(np.ascontiguousarray(image_hwc, np.float32) - w["average_rgb"]).permute(2, 0, 1)[None]
    for i in range(upto):
        k = torch.from_numpy(w["conv_kernel"][i]).permute(3, 2, 0, 1).contiguous()       # HWIO -> OIHW
        x = F.conv2d(x, k, torch.from_numpy(w["conv_bias"][i]), padding=1)
        if RELU[i]:
            x = F.relu(x)
        if POOL[i]:
            x = F.relu(F.max_pool2d(x, 2, 2))
    return x[0].permute(1, 2, 0).contiguous().numpy()


def netvlad(image_hwc, w):
    """The whole network: image -> unit-norm pca_dim vector (float32)."""
    x = torch.from_numpy(conv5_3(image_hwc, w))                      # [h, w, 512]
    x = x / torch.sqrt(torch.clamp((x * x).sum(-1, keepdim=True), min=1e-12))     # tf.nn.l2_normalize(dim=-1)
    P = x.shape[0] * x.shape[1]
    x = x.reshape(P, 512)
    a = torch.softmax(x @ torch.from_numpy(w["assignment"]), dim=-1)  # [P, K]
    C = torch.from_numpy(w["cluster_centers"])                        # [512, K]
    v = torch.einsum("pk,pdk->dk", a, x[:, :, None] + C[None])        # [512, K]
    v = v / torch.sqrt((v * v).sum(0, keepdim=True) + 1e-12)          # matconvnetNormalize over d, per cluster
    v = v.reshape(-1)                                                 # d-major, k-minor
    v = v / torch.sqrt((v * v).sum() + 1e-12)
    y = v @ torch.from_numpy(w["wpca_kernel"]) + torch.from_numpy(w["wpca_bias"])
    y = y / torch.sqrt(torch.clamp((y * y).sum(), min=1e-12))
    return y.numpy()
