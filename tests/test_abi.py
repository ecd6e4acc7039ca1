"""CPU: the C-ABI library loads, exports every symbol include/dh3d_hip.h declares, and rejects bad
arguments before touching the GPU (no compute calls here)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "dh3d_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(dh3d_[a-z0-9_]+)\s*\(", src)))


def test_header_and_binding_table_agree():
    from dh3d_amd import _lib
    assert sorted(_lib.EXPORTED_SYMBOLS) == declared_symbols()


def test_library_exports_every_declared_symbol():
    from dh3d_amd import _lib
    assert os.path.isfile(_lib.LIB_PATH), "build with `make -C dh3d_amd/csrc` or __graft_entry__.build()"
    handle = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared_symbols():
        assert hasattr(handle, name), name
    lib = _lib.lib()
    assert lib.dh3d_version() >= 100
    assert lib.dh3d_arch() == b"gfx950"
    assert lib.dh3d_status_string(0) == b"ok"


def test_invalid_arguments_are_status_codes_not_crashes():
    from dh3d_amd import _lib
    lib = _lib.lib()
    z = ctypes.c_void_p(0)
    assert lib.dh3d_knn_bruteforce(z, 1, 3, 8, 4, z, z, z) == 1          # null pointers
    one = ctypes.c_void_p(16)
    assert lib.dh3d_knn_bruteforce(one, 1, 17, 8, 4, one, one, z) == 2     # Dp > 16: beyond the staging tile
    assert lib.dh3d_knn_bruteforce(one, 1, 3, 8, 0, one, one, z) == 1      # K <= 0
    assert lib.dh3d_farthest_point_sample(1, 20000, 8, one, z, one, z) == 2  # N > 16384
    assert lib.dh3d_farthest_point_sample(1, 64, 0, one, z, one, z) == 1     # npoint <= 0 (tf_sampling.cpp:100)
    assert lib.dh3d_flex_conv_pm_fwd(one, one, one, one, 1, 64, 8, 48, 64, None, one, z) == 2
    assert lib.dh3d_netvlad_workspace_bytes(4, 1024, 128, 64) == 0
    assert lib.dh3d_netvlad_workspace_bytes(4, 1024, 256, 64) > 0


def test_library_was_built_from_this_tree():
    """The loaded library carries the hash of the sources it was compiled from (csrc/Makefile SRC_HASH -> dh3d_source_hash):
    a stale .so that travelled with the tree (built artefacts are git-ignored, not gpurun-ignored) fails here."""
    from dh3d_amd import _lib
    lib = _lib.lib()
    assert lib.dh3d_source_hash().decode() == _lib.tree_source_hash()


def test_python_ops_refuse_cpu_tensors():
    import pytest
    import torch
    from dh3d_amd import ops
    with pytest.raises(ValueError):
        ops.knn_bruteforce(torch.zeros(1, 3, 8), 4)  # CPU tensor: no fallback
    with pytest.raises(ValueError):
        ops.farthest_point_sample(4, torch.zeros(1, 8, 3))


def test_oracle_is_not_imported_by_the_product():
    pkg = os.path.join(ROOT, "dh3d_amd")
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            txt = open(os.path.join(pkg, fn)).read()
            assert "oracle" not in re.sub(r'""".*?"""', "", txt, flags=re.S), fn


def _code_only(path):
    """Source text without docstrings and comments."""
    import io
    import tokenize
    out = []
    for tok in tokenize.generate_tokens(io.StringIO(open(path).read()).readline):
        if tok.type == tokenize.COMMENT:
            continue
        if tok.type == tokenize.STRING and tok.string.lstrip("rbuRBU").startswith(('"""', "'''")):
            continue
        out.append(tok.string)
    return " ".join(out)


def test_training_module_has_no_tensor_op_restatement():
    """dh3d_amd/training.py runs the step on HIP kernels only: no tensor-op GEMM / activation / softmax path selectable at
    run time (the plain-torch restatement the HIP step is compared with lives in tests/torch_reference.py), and nothing in
    the package imports that test module."""
    import ast
    path = os.path.join(ROOT, "dh3d_amd", "training.py")
    tree = ast.parse(open(path).read())
    banned_attrs = {"matmul", "mm", "bmm", "addmm", "einsum", "softmax", "sigmoid", "relu", "linear", "batch_norm"}
    for node in ast.walk(tree):
        assert not (isinstance(node, ast.BinOp) and isinstance(node.op, ast.MatMult)), "matrix product at line %d" % node.lineno
        if isinstance(node, ast.Attribute) and isinstance(node.value, ast.Name) and node.value.id in ("torch", "F"):
            assert node.attr not in banned_attrs, "%s.%s at line %d" % (node.value.id, node.attr, node.lineno)
        if isinstance(node, (ast.Import, ast.ImportFrom)):
            names = [a.name for a in node.names] + [getattr(node, "module", "") or ""]
            assert not any("functional" in n for n in names), "torch.nn.functional imported at line %d" % node.lineno
        if isinstance(node, ast.arg):
            assert node.arg != "impl", "an implementation switch at line %d" % node.lineno
    pkg = os.path.join(ROOT, "dh3d_amd")
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            assert "torch_reference" not in _code_only(os.path.join(pkg, fn)), fn
