"""CPU-side checks of the C-ABI library (no GPU compute): it loads, exports every symbol that
include/ssdk.h declares, the host-only entry points are bit-exact with the reference fixtures, and
bad arguments come back as error codes with a message (never a crash)."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_header_symbols_exported():
    from ssds import _native as N

    hdr = open(os.path.join(ROOT, "include", "ssdk.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(ssdk_[a-z_0-9]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    assert declared == set(N.EXPORTS), declared ^ set(N.EXPORTS)
    for name in declared:
        assert hasattr(N.lib, name), name
    assert N.lib.ssdk_version() == 244 == N.ABI_VERSION


def test_descriptor_layouts_are_the_ones_the_library_was_built_with():
    """ssdk_struct_size (version 230): the ctypes mirrors of every descriptor struct have the size the library reports (the
    loader refuses to import otherwise), an unknown index reports 0, and the C compiler agrees with both about the header."""
    import subprocess
    import tempfile

    from ssds import _native as N

    classes = (N.Level, N.ConvDesc, N.MbConvDesc, N.FuseDesc, N.StemDesc, N.PoolDesc, N.XpairDesc, N.Op)
    sizes = [int(N.lib.ssdk_struct_size(i)) for i in range(len(classes))]
    assert sizes == [ctypes.sizeof(c) for c in classes] and all(sizes)
    assert N.lib.ssdk_struct_size(len(classes)) == 0 and N.lib.ssdk_struct_size(-1) == 0
    src = ('#include <stdio.h>\n#include "ssdk.h"\nint main(void){printf("%zu %zu %zu %zu %zu %zu %zu %zu\\n", sizeof(ssdk_level), '
           "sizeof(ssdk_conv_desc), sizeof(ssdk_mbconv_desc), sizeof(ssdk_fuse_desc), sizeof(ssdk_stem_desc), sizeof(ssdk_pool_desc), "
           "sizeof(ssdk_xpair_desc), sizeof(ssdk_op)); return 0;}\n")
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "s.c"), "w").write(src)
        subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), os.path.join(d, "s.c"), "-o", os.path.join(d, "s")], check=True)
        out = subprocess.run([os.path.join(d, "s")], check=True, capture_output=True, text=True).stdout.split()
    assert [int(v) for v in out] == sizes


def test_abi_check_rejects_another_header():
    """ssdk_abi_check (version 240): the library accepts the header it was built with (any last digit of the version) and
    rejects a caller compiled for another minor version or with another sizeof(ssdk_op) -- with a message, not a crash."""
    from ssds import _native as N

    sz = ctypes.sizeof(N.Op)
    assert N.lib.ssdk_abi_check(N.ABI_VERSION, sz) == 0
    assert N.lib.ssdk_abi_check(N.ABI_VERSION // 10 * 10 + 9, sz) == 0
    assert N.lib.ssdk_abi_check(N.ABI_VERSION - 10, sz) == -1 and b"ABI" in N.lib.ssdk_last_error()
    assert N.lib.ssdk_abi_check(N.ABI_VERSION, sz - 8) == -1 and b"sizeof" in N.lib.ssdk_last_error()


def test_library_is_in_tree_and_has_gfx950_code():
    from ssds import _native as N

    assert N.LIB_PATH.startswith(ROOT)
    blob = open(N.LIB_PATH, "rb").read()
    assert b"gfx950" in blob


def test_generate_anchors_bit_exact_with_reference_fixture(golden_dir):
    from ssds.modeling.layers import box

    g = np.load(os.path.join(golden_dir, "anchors.npz"))
    for k in g.files:
        if k.endswith("_spec"):
            continue
        spec = g[k + "_spec"]
        s, nr, ns = int(spec[0]), int(spec[1]), int(spec[2])
        r = [float(v) for v in spec[3:3 + nr]]
        sc = [float(v) for v in spec[3 + nr:3 + nr + ns]]
        got = box.generate_anchors(s, r, sc)
        assert got.dtype.is_floating_point and tuple(got.shape) == (nr * ns, 4)
        np.testing.assert_array_equal(got.numpy(), g[k], err_msg=k)


def test_bad_arguments_return_codes():
    from ssds import _native as N

    out = (ctypes.c_float * 4)()
    assert N.lib.ssdk_generate_anchors(0, out, 1, out, 1, out) == -1
    assert b"generate_anchors" in N.lib.ssdk_last_error()
    lv = N.Level()
    lv.A, lv.C, lv.H, lv.W, lv.stride = 3, 5, 7, 9, 8
    # top_n beyond the documented limit -> workspace query refuses (0) and decode returns BADARG
    assert N.lib.ssdk_decode_workspace_bytes(ctypes.byref(lv), 1, 2, 0, 5000) == 0
    assert N.lib.ssdk_decode(ctypes.byref(lv), 2, 0, 0.05, 5000, 1, None, None, None, None, 0, None) == -1
    assert N.lib.ssdk_decode_workspace_bytes(ctypes.byref(lv), 1, 2, 0, 50) > 0
    # null pointers
    assert N.lib.ssdk_nms(None, None, None, 1, 10, 0.5, 5, 1, None, None, None, None, 0, None) == -1
    assert N.lib.ssdk_match_targets(None, 1, 1, out, 1, 1, 1, 1, 8, 0.5, 0.4, 0.0, None, None, None, None) == -1


def test_no_cpu_fallback():
    import torch
    from ssds import _native as N
    from ssds.modeling.layers import box

    cls = torch.zeros(1, 2, 2, 2)
    loc = torch.zeros(1, 4, 2, 2)
    anc = torch.tensor([[-4.0, -4, 11, 11]])
    with pytest.raises(N.SsdkError, match="no CPU fallback"):
        box.decode(cls, loc, 8, 0.05, 10, anc)
    with pytest.raises(N.SsdkError, match="no CPU fallback"):
        box.nms(torch.zeros(1, 4), torch.zeros(1, 4, 4), torch.zeros(1, 4))


def test_no_cpu_fallback_for_training_and_eval_kernels():
    """The fused loss, the mAP bookkeeping, target assignment and the fused conv entry points refuse host tensors."""
    from collections import OrderedDict

    import torch
    from ssds import _native as N
    from ssds.core.evaluation_metrics import MeanAveragePrecision
    from ssds.core.fused_loss import match_loss
    from ssds.modeling.layers import box

    anchors = OrderedDict([(8, torch.tensor([[-4.0, -4, 11, 11]]))])
    targets = torch.tensor([[[1.0, 1, 8, 8, 0]]])
    with pytest.raises(N.SsdkError, match="no CPU fallback"):
        match_loss(torch.zeros(1, 3, 2, 2), torch.zeros(1, 4, 2, 2), targets, anchors, 3, 8, [0.5, 0.4])
    with pytest.raises(N.SsdkError, match="no CPU fallback"):
        box.extract_targets(targets, anchors, 3, 8, (2, 2), [0.5, 0.4])
    with pytest.raises(N.SsdkError, match="no CPU fallback"):
        MeanAveragePrecision(3, 0.1, 0.5)((torch.zeros(1, 4), torch.zeros(1, 4, 4), torch.zeros(1, 4)), targets)
    # bad arguments come back as error codes, not crashes
    assert N.lib.ssdk_match_loss(None, 1, 1, None, 1, 1, 1, 1, 8, 0, 0.5, 0.4, 0.0, None, None, 0, 0.25, 2.0, 0.11, 0,
                                 None, None, None, None, 0, None) == -1
    assert N.lib.ssdk_map_match(None, None, None, 1, 4, None, 1, 3, 0.1, 0.5, None, None, None, None) == -1
    assert N.lib.ssdk_map_average_precision(None, None, None, 3, None, None) == -1
    assert N.lib.ssdk_match_loss_workspace_bytes(0, 1, 1, 1) == 0
