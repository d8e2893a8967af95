"""CPU checks of the boundary: the C-ABI library loads, exports every symbol include/mgen.h declares, the ctypes
table covers exactly those symbols, and argument validation works without a GPU."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    txt = open(os.path.join(ROOT, "include", "mgen.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(mgen_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    from mgen import _lib
    assert os.path.exists(_lib.LIB_PATH), "build the library first: make -C symbolic-music-generation_amd/csrc"
    l = ctypes.CDLL(_lib.LIB_PATH)
    names = header_functions()
    assert len(names) >= 25
    for n in names:
        assert hasattr(l, n), f"{n} declared in include/mgen.h but not exported"


def test_ctypes_table_matches_header():
    from mgen import _lib
    declared = set(header_functions())
    bound = set(_lib.SIGNATURES) | {"mgen_last_error", "mgen_version", "mgen_source_id", "mgen_sumsq_partial_count", "mgen_device_info",
                                     "mgen_attn_decode_workspace", "mgen_gemm_slab_count", "mgen_gemm_skinny_fits", "mgen_gemm_skinny_slabs", "mgen_lmhead_ce_workspace_bytes", "mgen_decode_layer_fits"}
    assert declared == bound, (declared - bound, bound - declared)


def test_header_arity_matches_ctypes_table():
    from mgen import _lib
    txt = open(os.path.join(ROOT, "include", "mgen.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    for name, args in _lib.SIGNATURES.items():
        m = re.search(r"\b" + name + r"\s*\((.*?)\)\s*;", txt, flags=re.S)
        assert m, name
        n = len([a for a in m.group(1).split(",") if a.strip() and a.strip() != "void"])
        assert n == len(args), (name, n, len(args))


def test_argument_validation_needs_no_gpu():
    from mgen import _lib
    l = _lib.lib()
    assert l.mgen_version() >= 100
    # null pointers / bad shapes are rejected before any launch
    rc = l.mgen_gemm_bf16(None, None, None, None, 8, 8, 8, 8, 8, 8, 0, 0, 0, 0, 1.0, None)
    assert rc == -1 and b"null" in l.mgen_last_error()
    rc = l.mgen_rmsnorm_fwd(8, 8, 8, None, 4, 510, 1e-6, 1, None)
    assert rc == -1 and b"d" in l.mgen_last_error()
    with pytest.raises(_lib.MgenError):
        _lib.call("mgen_adamw_step", 16, 16, 16, 16, None, 10, None, 1e-3, 0.9, 0.95, 1e-5, 0.1, 1, None)


def test_library_was_built_from_the_sources_beside_it():
    """A library left over from older sources (a failed build) must not pass for the current one: the loader compares
    the source identity compiled into the library with the one of csrc/ and refuses a mismatch."""
    from mgen import _lib
    from mgen._srcid import source_id
    l = _lib.lib()
    csrc = os.path.join(ROOT, "symbolic-music-generation_amd", "csrc")
    if not os.environ.get("MGEN_LIB"):
        assert l.mgen_source_id().decode() == source_id(csrc)

    class Stale:
        @staticmethod
        def mgen_source_id():
            return b"0123456789abcdef"
    if not os.environ.get("MGEN_LIB"):
        with pytest.raises(_lib.MgenError, match="other sources"):
            _lib._check_source_id(Stale)


def test_few_row_projection_shape_rules_need_no_gpu():
    """Host-side shape rules of the decode-step projections: the slab count gives every CU of a 256-CU chip a
    workgroup with as few slabs as that takes and always divides the number of 512-index chunks; the layer call takes
    the default decoder and the 650M model at 128 streams."""
    from mgen import _lib
    l = _lib.lib()
    for M, N, K in [(128, 512, 1024), (128, 512, 2048), (128, 1536, 1536), (128, 1536, 6144), (256, 512, 2048),
                    (1, 512, 1024), (1000, 1536, 6144)]:
        s = l.mgen_gemm_skinny_slabs(M, N, K)
        chunks = K // 512
        tiles = ((M + 31) // 32) * (N // 64)
        assert 1 <= s <= chunks and chunks % s == 0
        assert s == chunks or tiles * s >= 256
        assert all(tiles * d < 256 for d in range(1, s) if chunks % d == 0)
        assert l.mgen_gemm_skinny_fits(M, N, K, K, K, N, s) == 1
    assert l.mgen_gemm_skinny_slabs(128, 512, 1024) == 2 and l.mgen_gemm_skinny_slabs(128, 512, 2048) == 4
    assert l.mgen_gemm_skinny_slabs(128, 1536, 1536) == 3 and l.mgen_gemm_skinny_slabs(128, 1536, 6144) == 3
    assert l.mgen_gemm_skinny_fits(128, 4608, 1536, 1536, 1536, 4608, 0) == 1    # bf16 output over three chunks
    assert l.mgen_gemm_skinny_fits(128, 512, 1536, 1536, 1536, 512, 2) == 0      # 2 slabs do not divide 3 chunks
    assert l.mgen_decode_layer_fits(128, 512, 8, 8, 128, 2048) == 1
    assert l.mgen_decode_layer_fits(128, 1536, 24, 24, 64, 6144) == 1
    assert l.mgen_decode_layer_fits(128, 512, 8, 8, 128, 2000) == 0
