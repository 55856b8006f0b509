"""Error behaviour at the C ABI (include/p2pb_hip.h: "0 on success, P2PB_EINVAL (-22) for unsupported arguments, else the
hipError_t of the failed launch"; the reference exit(-1)s on a failed launch, PN2/cuda_utils.cuh:30-39, and its Python
side raises RuntimeError on non-CUDA / non-contiguous / wrong-dtype tensors, PN2/utils.hpp:7-18): every entry point
rejects degenerate sizes and missing workspaces with -22 WITHOUT launching anything, the ctypes layer turns that into
P2PBError (a RuntimeError), the tensor preconditions raise RuntimeError -- and the device is still healthy afterwards."""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu
_i, _f = ctypes.c_int, ctypes.c_float
NULL = ctypes.c_void_p(0)


@pytest.fixture(scope="module")
def L():
    from p2p_bridge_amd import _lib
    return _lib


def test_degenerate_sizes_are_einval(L):
    lib = L.lib()
    x = torch.zeros(64, device="cuda")
    p = L.ptr(x)
    s = L.stream_ptr()
    calls = {
        "p2pb_furthest_point_sampling": (_i(0), _i(8), _i(2), p, NULL, p, s),
        "p2pb_furthest_point_sampling_grid": (_i(1), _i(8), _i(2), p, NULL, p, s),          # no workspace
        "p2pb_furthest_point_sampling_coop": (_i(1), _i(100), _i(2), p, p, p, s),             # n <= 16384
        "p2pb_ball_query": (_i(1), _i(0), _i(4), _f(0.1), _i(8), p, p, p, s),
        "p2pb_chamfer_forward": (_i(0), _i(4), _i(4), p, p, p, p, p, p, s),
        "p2pb_conv3d_k3_forward": (_i(1), _i(0), _i(8), _i(8), p, p, p, NULL, NULL, _i(0), p, NULL, s),
        "p2pb_conv3d_k3_forward_ex": (_i(1), _i(8), _i(8), _i(5), p, p, p, NULL, NULL, NULL, _i(0), NULL, _i(4), p, NULL, s),  # r = 5
        "p2pb_pointwise_conv_forward": (_i(1), _i(8), _i(8), _i(0), p, p, p, NULL, NULL, NULL, _i(0), _i(0), p, NULL, s),
        "p2pb_gn_affine_params": (_i(1), _i(12), _i(5), _i(1), ctypes.c_double(4.0), p, NULL, NULL, NULL, _i(0), _f(1e-5), p, p, NULL, s),  # 12 % 5
        "p2pb_radius_count": (_i(1), _i(4), p, p, _f(-1.0), p, s),                             # negative radius
        "p2pb_linear_attention_forward": (_i(1), _i(2), _i(16), _i(8), p, p, NULL, s),        # dim_head != 32
        "p2pb_knn_points": (_i(1), _i(1), _i(4), _i(9), p, p, p, p, NULL, p, s),              # k > n
        "p2pb_three_nn": (_i(1), _i(0), _i(4), p, p, p, p, s),
        # round 6: the training step's folded entry points
        "p2pb_affine_act_train": (_i(1), _i(8), _i(8), p, p, p, _i(1), NULL, NULL, _f(0.1), NULL, ctypes.c_uint(1), p, s),  # dropout without a seed
        "p2pb_grouping_backward_pitched": (_i(1), _i(2), _i(8), _i(2), _i(2), p, ctypes.c_long(7), p, p, s),           # pitch < c*m*u
        "p2pb_three_nn_interpolate_backward_pitched": (_i(1), _i(2), _i(8), _i(4), p, ctypes.c_long(15), p, p, p, s),  # pitch < c*n
    }
    nab = lambda **k: (_i(1), _i(8), _i(2), _i(8), p, p, p, p, p, NULL, NULL, NULL, _i(0), _i(k.get("swish", 0)),  # noqa: E731
                       ctypes.c_long(k.get("pitch", 0)), k.get("gmean", NULL), k.get("res", NULL), k.get("rgate", NULL),
                       _f(k.get("drop", 0.0)), k.get("seed", NULL), ctypes.c_uint(0), p, NULL, NULL, NULL, k.get("dres", NULL),
                       k.get("drgate", NULL), p, s)
    for bad in (dict(drop=1.0, seed=p), dict(drop=0.2), dict(gmean=p, swish=1), dict(gmean=p, drop=0.1, seed=p), dict(pitch=63),
                dict(pitch=66), dict(dres=p, drgate=p), dict(res=p, rgate=p, dres=p)):
        assert lib.p2pb_norm_act_backward_ex(*nab(**bad)) == -22, bad
    for name, args in calls.items():
        rc = getattr(lib, name)(*args)
        assert rc == -22, (name, rc)
    with pytest.raises(L.P2PBError):
        L.call("p2pb_furthest_point_sampling", _i(0), _i(8), _i(2), p, NULL, p, s)
    assert issubclass(L.P2PBError, RuntimeError)
    torch.cuda.synchronize()  # nothing was launched, nothing is pending
    assert float((torch.ones(4, device="cuda") * 2).sum()) == 8.0


def test_tensor_preconditions_raise_runtime_error():
    from p2p_bridge_amd import pointnet2_batch_cuda as ext

    c = torch.rand(2, 3, 64, device="cuda")
    with pytest.raises(RuntimeError, match="CUDA"):
        ext.furthest_point_sampling_forward(c.cpu(), 8)
    with pytest.raises(RuntimeError, match="contiguous"):
        ext.furthest_point_sampling_forward(c.transpose(1, 2).contiguous().transpose(1, 2), 8)
    with pytest.raises(RuntimeError, match="float"):
        ext.furthest_point_sampling_forward(c.double(), 8)
    assert ext.furthest_point_sampling_forward(c, 8).shape == (2, 8)
