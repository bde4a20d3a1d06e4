"""ctypes binding of libtgcn_hip.so (include/tgcn_hip.h).  There is no CPU fallback: if the shared
library is missing or a tensor is not on a ROCm device the call fails loudly."""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(_HERE)
LIB_PATH = os.path.join(_HERE, "lib", "libtgcn_hip.so")
SOURCES = [os.path.join(_HERE, "csrc", "tgcn_hip.hip")]      # one translation unit; the kernels are in csrc/*.h
HEADERS = [os.path.join(_HERE, "csrc", h) for h in ("common.h", "hop.h", "project.h", "wgrad.h", "small_graph.h", "pool_relayout.h", "device_build.h", "graph_build.h")]
INCLUDE = os.path.join(ROOT, "include")


class TgcnError(RuntimeError):
    pass


class CsrStruct(C.Structure):
    _fields_ = [("n", C.c_int64), ("nnz", C.c_int64), ("rowptr", C.c_void_p), ("edges", C.c_void_p), ("dense", C.c_void_p)]


class SchedStruct(C.Structure):
    _fields_ = [("lanes_per_row", C.c_int32), ("row_thresh", C.c_int32), ("nblk", C.c_int32), ("nseg", C.c_int32),
                ("nlong", C.c_int32), ("nhuge", C.c_int32), ("npartial", C.c_int32), ("seg_mode", C.c_int32), ("row_mix", C.c_int32), ("nwseg", C.c_int32),
                ("blk_row", C.c_void_p), ("seg_row", C.c_void_p), ("seg_e0", C.c_void_p), ("seg_e1", C.c_void_p),
                ("seg_slot", C.c_void_p), ("long_row", C.c_void_p), ("long_slot", C.c_void_p)]


class DenseStruct(C.Structure):
    _fields_ = [("ptr", C.c_void_p), ("batch_stride", C.c_int64), ("row_stride", C.c_int64)]


# name -> (restype, argtypes); must list every symbol include/tgcn_hip.h declares
_P = C.c_void_p
SIGNATURES = {
    "tgcn_last_error": (C.c_char_p, []),
    "tgcn_abi_version": (C.c_int, []),
    "tgcn_graph_create_from_coo": (C.c_int, [C.c_int64, C.c_int64, C.c_int64, _P, _P, _P, C.POINTER(_P)]),
    "tgcn_graph_create_from_csr": (C.c_int, [C.c_int64, C.c_int64, _P, _P, _P, C.POINTER(_P)]),
    "tgcn_graph_create_from_edge_index": (C.c_int, [C.c_int64, C.c_int64, _P, _P, C.POINTER(_P)]),
    "tgcn_graph_csr": (C.POINTER(CsrStruct), [_P]),
    "tgcn_graph_n_cols": (C.c_int64, [_P]),
    "tgcn_graph_destroy": (None, [_P]),
    "tgcn_graclus_match_f32": (C.c_int, [C.c_int64, _P, _P, _P, C.c_int64, _P, _P, _P]),
    "tgcn_graclus_match_f64": (C.c_int, [C.c_int64, _P, _P, _P, C.c_int64, _P, _P, _P]),
    "tgcn_sched_build": (C.c_int, [_P, C.c_int32, C.c_int, C.POINTER(_P)]),
    "tgcn_sched_build_csr": (C.c_int, [C.POINTER(CsrStruct), C.c_int64, C.c_int32, C.c_int, C.POINTER(_P)]),
    "tgcn_sched_copy": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "tgcn_csr_build_workspace_bytes": (C.c_size_t, [C.c_int64, C.c_int64]),
    "tgcn_csr_build_f32": (C.c_int, [_P, C.c_int64, C.c_int64, C.c_int64, _P, _P, _P, _P, _P, _P, C.c_size_t]),
    "tgcn_sched_get": (C.POINTER(SchedStruct), [_P]),
    "tgcn_sched_destroy": (None, [_P]),
    "tgcn_profile_start": (C.c_int, [C.c_int32]),
    "tgcn_profile_stop": (C.c_int, [C.POINTER(C.c_int32), C.POINTER(C.c_float), C.c_int32, C.POINTER(C.c_int32)]),
    "tgcn_set_tuning": (C.c_int, [C.c_char_p, C.c_int32]),
    "tgcn_reset_tuning": (None, []),
    "tgcn_csr_sddmm_f32": (C.c_int, [_P, C.POINTER(CsrStruct), C.c_int64, C.c_int32, C.c_int32, C.POINTER(DenseStruct), C.POINTER(DenseStruct), C.c_float, _P,
                                     C.c_int32]),
    "tgcn_cheb_project_mapped_f32": (C.c_int, [_P, C.c_int64, C.c_int32, C.c_int32, C.c_int32, _P, _P, _P, _P, C.c_int32, C.c_int64, C.c_int64, _P, C.c_uint32,
                                                C.c_int32, _P, C.c_int64, _P, C.c_int64]),
    "tgcn_edge_normalise_workspace_bytes": (C.c_size_t, [C.c_int64, C.c_int64]),
    "tgcn_edge_normalise_f32": (C.c_int, [_P, C.c_int64, C.c_int64, _P, _P, _P, _P, _P, C.POINTER(C.c_int64), _P, C.c_size_t]),
    "tgcn_adjacency_normalise_workspace_bytes": (C.c_size_t, [C.c_int64, C.c_int64]),
    "tgcn_adjacency_normalise_f32": (C.c_int, [_P, C.c_int64, C.c_int64, _P, _P, _P, C.c_float, _P, _P, _P, C.POINTER(C.c_int64), _P, C.c_size_t]),
    "tgcn_hop_vec_width": (C.c_int, [C.c_int32, C.c_int]),
    "tgcn_hop_lanes_per_row": (C.c_int, [C.c_int32, C.c_int]),
    "tgcn_hop_groups_per_block": (C.c_int, [C.c_int32, C.c_int]),
    "tgcn_csr_hop_workspace_bytes": (C.c_size_t, [C.POINTER(SchedStruct), C.c_int32, C.c_int32, C.c_int]),
    "tgcn_csr_hop_f32": (C.c_int, [_P, C.POINTER(CsrStruct), C.POINTER(SchedStruct), C.c_int32, C.c_int32,
                                   C.POINTER(DenseStruct), C.POINTER(DenseStruct), C.c_float, C.c_float,
                                   C.POINTER(DenseStruct), C.POINTER(DenseStruct), _P, C.c_size_t]),
    "tgcn_csr_hop2_f32": (C.c_int, [_P, C.POINTER(CsrStruct), C.POINTER(SchedStruct), C.c_int32, C.c_int32,
                                    C.POINTER(DenseStruct), C.POINTER(DenseStruct), C.c_float, C.c_float,
                                    C.POINTER(DenseStruct), C.c_float, C.POINTER(DenseStruct), C.POINTER(DenseStruct), _P, C.c_size_t]),
    "tgcn_cheb_forward_pf_workspace_bytes": (C.c_size_t, [C.POINTER(SchedStruct), C.c_int32, C.c_int64, C.c_int64, C.c_int32]),
    "tgcn_cheb_forward_pf_f32": (C.c_int, [_P, C.POINTER(CsrStruct), C.POINTER(SchedStruct), C.c_int32, C.c_int32, C.c_int64,
                                           C.c_int64, C.c_int32, C.c_int32, _P, _P, _P, C.c_int32, _P, _P, C.c_size_t]),
    "tgcn_cheb_project_first_f32": (C.c_int, [_P, C.c_int64, C.c_int64, C.c_int32, C.c_int32, C.c_int32, _P, _P, _P, C.c_int32, _P, _P]),
    "tgcn_cheb_project_f32": (C.c_int, [_P, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.POINTER(_P),
                                        C.POINTER(C.c_int64), _P, _P, C.c_int32, C.c_int64, C.c_int64, C.c_int32,
                                        _P, C.c_int64]),
    "tgcn_cheb_project_windows_f32": (C.c_int, [_P, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.POINTER(_P), _P, _P,
                                                C.c_int32, _P]),
    "tgcn_cheb_wgrad_workspace_bytes": (C.c_size_t, [C.c_int64, C.c_int32, C.c_int32, C.c_int32]),
    "tgcn_cheb_wgrad_f32": (C.c_int, [_P, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.POINTER(_P), C.POINTER(C.c_int64), _P,
                                      C.c_int64, _P, _P, C.c_size_t]),
    "tgcn_relayout_qnc_to_nqc_f32": (C.c_int, [_P, _P, _P, C.c_int64, C.c_int64, C.c_int32]),
    "tgcn_cheb_forward_workspace_bytes": (C.c_size_t, [C.POINTER(SchedStruct), C.c_int32, C.c_int64, C.c_int64,
                                                       C.c_int32, C.c_int32, C.c_int64]),
    "tgcn_cheb_forward_f32": (C.c_int, [_P, C.POINTER(CsrStruct), C.POINTER(SchedStruct), C.c_int32, C.c_int32,
                                        C.c_int64, C.c_int64, C.c_int32, C.c_int32, _P, _P, _P, C.c_int32, _P,
                                        C.c_int32, C.c_int64, _P, C.c_size_t]),
    "tgcn_cheb_forward_pool_workspace_bytes": (C.c_size_t, [C.POINTER(SchedStruct), C.c_int32, C.c_int64, C.c_int64, C.c_int32, C.c_int32,
                                                            C.c_int32, C.c_int64, C.c_int32]),
    "tgcn_cheb_forward_pool_f32": (C.c_int, [_P, C.POINTER(CsrStruct), C.POINTER(SchedStruct), C.c_int32, C.c_int32, C.c_int64, C.c_int64,
                                             C.c_int32, C.c_int32, _P, _P, _P, C.c_int32, C.c_int32, _P, _P, C.c_int32, C.c_int64, _P,
                                             C.c_size_t]),
    "tgcn_cheb_forward_compact_workspace_bytes": (C.c_size_t, [C.POINTER(SchedStruct), C.c_int32, C.c_int64, C.c_int64, C.c_int32, C.c_int64]),
    "tgcn_cheb_forward_compact_f32": (C.c_int, [_P, C.POINTER(CsrStruct), C.POINTER(CsrStruct), C.POINTER(SchedStruct), C.c_int32,
                                                C.c_int64, C.c_int64, C.c_int32, C.c_int32, _P, _P, _P, C.c_int32, _P, _P, _P,
                                                C.c_int64, _P, C.c_int64, _P, C.c_size_t]),
    "tgcn_cheb_compact_layer_workspace_bytes": (C.c_size_t, [C.POINTER(SchedStruct), C.c_int32, C.c_int32, C.c_int64, C.c_int64, C.c_int32, C.c_int64, C.c_int32]),
    "tgcn_cheb_compact_layer_f32": (C.c_int, [_P, C.POINTER(CsrStruct), C.POINTER(CsrStruct), C.POINTER(SchedStruct), C.c_int32, C.c_int32,
                                              C.c_int64, C.c_int64, C.c_int32, C.c_int32, _P, _P, _P, _P, C.c_int32, _P, _P, _P,
                                              C.c_int64, _P, C.c_int64, _P, _P, C.c_size_t]),
    "tgcn_cheb_forward_small_supported": (C.c_int, [C.c_int64, C.c_int64, C.c_int32, C.c_int32]),
    "tgcn_cheb_basis_small_supported": (C.c_int, [C.c_int64, C.c_int64, C.c_int32, C.c_int32]),
    "tgcn_cheb_forward_small_pool_supported": (C.c_int, [C.c_int64, C.c_int64, C.c_int32, C.c_int32]),
    "tgcn_cheb_basis_small_f32": (C.c_int, [C.c_void_p, C.POINTER(CsrStruct), C.c_int32, C.c_int32, C.c_int64, C.c_int32,
                                           C.c_void_p, C.c_void_p]),
    "tgcn_cheb_forward_small_f32": (C.c_int, [_P, C.POINTER(CsrStruct), C.c_int32, C.c_int32, C.c_int64, C.c_int32, C.c_int32,
                                              _P, _P, _P, _P, C.c_int32, _P]),
    "tgcn_cheb_forward_small_pool_f32": (C.c_int, [_P, C.POINTER(CsrStruct), C.c_int32, C.c_int32, C.c_int64, C.c_int32, C.c_int32,
                                                   _P, _P, _P, _P, C.c_int32, C.c_int32, C.c_int32, _P, _P]),
    "tgcn_relu_pool_f32": (C.c_int, [_P, _P, _P, _P, C.c_int64, C.c_int64, C.c_int32, C.c_int32]),
    "tgcn_relu_pool_bwd_f32": (C.c_int, [_P, _P, _P, _P, _P, C.c_int64, C.c_int64, C.c_int32, C.c_int32]),
    "tgcn_cheb_windows_wgrad_workspace_bytes": (C.c_size_t, [C.c_int64, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
    "tgcn_cheb_windows_backward_f32": (C.c_int, [_P, C.c_int64, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _P, _P, _P, _P, _P,
                                                 _P, C.c_size_t]),
    "tgcn_fold_weight_f32": (C.c_int, [_P, C.c_int32, C.c_int64, _P, _P, _P, C.c_int32]),
    "tgcn_weight_layout_f32": (C.c_int, [_P, C.c_int32, C.c_int32, C.c_int32, _P, _P, C.c_int32]),
    "tgcn_csr_hop_f64": (C.c_int, [_P, C.c_int64, _P, _P, _P, C.c_int64, _P, _P, C.c_double, C.c_double, _P, _P]),
    "tgcn_pack_rows_f32": (C.c_int, [_P, _P, C.c_int64, _P, C.c_int64, C.c_int32, _P]),
    "tgcn_pool_max_f32": (C.c_int, [_P, _P, _P, _P, C.c_int64, C.c_int64, C.c_int32, C.c_int32]),
    "tgcn_pool_max_bwd_f32": (C.c_int, [_P, _P, _P, _P, C.c_int64, C.c_int64, C.c_int32, C.c_int32]),
}

_lib = None
ABI_VERSION = 7      # include/tgcn_hip.h: TGCN_ABI_VERSION


def source_hash():
    """sha256 over the HIP sources and the header: names the code a binary or a committed counter file belongs to"""
    import hashlib
    h = hashlib.sha256()
    for path in SOURCES + HEADERS + [os.path.join(INCLUDE, "tgcn_hip.h")]:
        with open(path, "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def binary_hash():
    """source hash recorded next to the library when it was built (LIB_PATH.srchash): names the code of the BINARY that lib()
    loads, which is what a committed counter file has to match -- the sources on disk may have moved on"""
    try:
        return open(LIB_PATH + ".srchash").read().strip()
    except OSError:
        return None


def build(verbose=False, force=False):
    """Compile the HIP sources for gfx950 into tgcn_amd/lib/libtgcn_hip.so (hipcc cross-compiles without a GPU).
    force=False reuses a library whose recorded source hash matches the sources (modification times say nothing about a
    binary that travelled with the tree); __graft_entry__.build() passes force=True, so the recipe itself is exercised."""
    os.makedirs(os.path.dirname(LIB_PATH), exist_ok=True)
    stamp = LIB_PATH + ".srchash"
    want = source_hash()
    if not force and os.path.exists(LIB_PATH) and os.path.exists(stamp) and open(stamp).read().strip() == want:
        return LIB_PATH
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    tmp = LIB_PATH + ".%d.tmp" % os.getpid()      # per process: concurrent builders must not share a half-written file
    cmd = [hipcc, "-O3", "--offload-arch=gfx950", "-std=c++17", "-shared", "-fPIC", "-I", INCLUDE, "-o", tmp] + SOURCES
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    try:        # the stamp certifies a library that loads and speaks this ABI, not merely one that linked
        probe = C.CDLL(tmp)
        probe.tgcn_abi_version.restype = C.c_int
        if probe.tgcn_abi_version() != ABI_VERSION:
            raise TgcnError("tgcn_amd: freshly built library reports ABI %d, expected %d" % (probe.tgcn_abi_version(), ABI_VERSION))
    except OSError as e:
        os.unlink(tmp)
        raise TgcnError("tgcn_amd: freshly built library does not load: %s" % e)
    os.replace(tmp, LIB_PATH)
    with open(stamp, "w") as f:
        f.write(want + "\n")
    return LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise TgcnError("tgcn_amd: %s is missing -- run `python -c 'import __graft_entry__ as g; g.build()'` "
                            "(there is no CPU fallback)" % LIB_PATH)
        handle = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)
            fn.restype, fn.argtypes = res, args
        if handle.tgcn_abi_version() != ABI_VERSION:
            raise TgcnError("tgcn_amd: ABI version mismatch")
        _lib = handle
    return _lib


def loaded():
    """True once lib() has mapped the library (test teardown resets the tuning switches only then)."""
    return _lib is not None


def check(rc):
    if rc != 0:
        raise TgcnError("libtgcn_hip: %s (code %d)" % (lib().tgcn_last_error().decode(), rc))


def require_device(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise TgcnError("tgcn_amd: the HIP path needs tensors on a ROCm device (got %s); there is no CPU fallback" % t.device)


def stream_ptr():
    """Raw handle of torch's current stream on the current device (the private accessor skips building a Stream
    object: 1 us instead of 25 us per call on the latency-bound small-graph path)."""
    import torch
    raw = getattr(torch._C, "_cuda_getCurrentRawStream", None)
    if raw is not None:
        return C.c_void_p(raw(torch.cuda.current_device()))
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def profile_start(capacity=4096):
    check(lib().tgcn_profile_start(capacity))


def profile_stop(capacity=4096):
    """-> list of (kind, ms) per launch, in launch order (kinds: include/tgcn_hip.h TGCN_PROF_*)."""
    kinds = (C.c_int32 * capacity)()
    ms = (C.c_float * capacity)()
    n = C.c_int32(0)
    check(lib().tgcn_profile_stop(kinds, ms, capacity, C.byref(n)))
    return [(kinds[i], ms[i]) for i in range(n.value)]
