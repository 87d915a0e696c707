"""GPU: NetVLAD descriptor inference (csrc/k_cnn.hip) against the PyTorch fp32 CPU
evaluation of the same published network (oracle/netvlad_torch.py).  fp32 matrix-core products are exact and
accumulate in fp32, so the two differ by summation order only: tolerance 1e-4 absolute on the unit-norm descriptor
(the NN stage thresholds distances at 0.13), 2e-3 relative on the raw trunk activations."""
import numpy as np
import pytest

from multi_robot_slam_separators_amd import lib, synth
from oracle import netvlad_torch as ref

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def model():
    import torch
    w = ref.random_weights(3, clusters=64, pca_dim=512)
    f = lib.SeparatorFinder(synth.camera_params(), device=0)
    f.set_stream(torch.cuda.current_stream().cuda_stream)
    f.netvlad_load(w)
    yield f, w
    f.close()


def infer(f, torch, image, n_out):
    dev = torch.device("cuda:0")
    d_img = torch.from_numpy(np.ascontiguousarray(image, np.float32)).to(dev)
    d_out = torch.zeros(n_out, dtype=torch.float32, device=dev)
    f.netvlad_infer_device(d_img.data_ptr(), image.shape[1], image.shape[0], d_out.data_ptr(), n_out)
    torch.cuda.synchronize()
    return d_out.cpu().numpy()


@pytest.mark.parametrize("shape", [(64, 96), (120, 160), (50, 70), (240, 320), (480, 752)])
def test_descriptor_equals_torch_reference(model, shape):
    import torch
    f, w = model
    rng = np.random.default_rng(shape[0])
    image = rng.uniform(0, 255, size=(shape[0], shape[1], 3)).astype(np.float32)
    got = infer(f, torch, image, 512)
    want = ref.netvlad(image, w)
    assert abs(float(np.linalg.norm(got)) - 1.0) < 1e-5
    assert np.max(np.abs(got - want)) < 1e-4, float(np.max(np.abs(got - want)))
    # the prefix the reference keeps (data_handler.py:157-158) is the same prefix
    assert np.array_equal(infer(f, torch, image, 128), got[:128])


def test_distinct_images_give_distinct_descriptors_and_errors(model):
    import torch
    f, w = model
    rng = np.random.default_rng(1)
    a = rng.uniform(0, 255, size=(64, 96, 3)).astype(np.float32)
    b = a + rng.normal(0, 2.0, size=a.shape).astype(np.float32)
    c = rng.uniform(0, 255, size=(64, 96, 3)).astype(np.float32)
    da, db, dc = infer(f, torch, a, 512), infer(f, torch, b, 512), infer(f, torch, c, 512)
    assert np.linalg.norm(da - db) < np.linalg.norm(da - dc)
    with pytest.raises(lib.SepfinderError):
        infer(f, torch, a[:8, :8], 16)                       # smaller than the four poolings need
    with pytest.raises(lib.SepfinderError):
        infer(f, torch, a, 4096)                             # more dimensions than the loaded WPCA has


def test_batch_equals_single_images(model):
    """sf_netvlad_infer_batch_device (data_handler.py:149-156: netvlad_batch_size = 3 images per call): every image's
    descriptor carries the bits of its single-image call, for batches that fill one WPCA group and that span two."""
    import torch
    f, w = model
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(11)
    # heights that are a multiple of 16 run the trunk of a group as ONE vertical stack of its images (a tap must not see
    # the neighbouring image's rows); the others image by image
    for height, width, sizes in ((120, 160, (3, 5)), (128, 160, (2, 3, 4, 5)), (96, 64, (3,)), (16, 48, (4,))):
        for n_img in sizes:
            imgs = rng.uniform(0, 255, size=(n_img, height, width, 3)).astype(np.float32)
            d_imgs = torch.from_numpy(imgs).to(dev)
            d_out = torch.zeros((n_img, 128), dtype=torch.float32, device=dev)
            f.netvlad_infer_batch_device(d_imgs.data_ptr(), n_img, width, height, d_out.data_ptr(), 128)
            torch.cuda.synchronize()
            got = d_out.cpu().numpy()
            for i in range(n_img):
                assert np.array_equal(got[i], infer(f, torch, imgs[i], 128)), (height, width, n_img, i)


def test_published_wpca_width_at_camera_resolution():
    """The network as the reference loads it -- 64 clusters, 4096-wide WPCA (data_handler.py:63 vgg16NetvladPca) -- on a
    640 x 480 image, against the PyTorch fp32 CPU evaluation: 1e-4 on the unit-norm descriptor, and the 128-value
    prefix the reference keeps (data_handler.py:157-158)."""
    import torch
    w = ref.random_weights(5, clusters=64, pca_dim=4096)
    rng = np.random.default_rng(12)
    image = rng.uniform(0, 255, size=(480, 640, 3)).astype(np.float32)
    with lib.SeparatorFinder(synth.camera_params(), device=0) as f:
        f.set_stream(torch.cuda.current_stream().cuda_stream)
        f.netvlad_load(w)
        got = infer(f, torch, image, 4096)
        pre = infer(f, torch, image, 128)
    want = ref.netvlad(image, w)
    assert got.shape == (4096,) and abs(float(np.linalg.norm(got)) - 1.0) < 1e-5
    assert np.max(np.abs(got - want)) < 1e-4, float(np.max(np.abs(got - want)))
    assert np.array_equal(pre, got[:128])
