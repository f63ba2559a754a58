"""ctypes binding of libplvs_hip.so (the C ABI declared in include/plvs_hip.h).

The library holds the hand-written gfx950 kernels; there is no Python/CPU
fallback: if it is missing or fails to load, importing the product path raises.
"""
import ctypes
import os

# torch must be imported before the library is dlopen()ed: both need
# libamdhip64.so.7 and the loader then shares torch's copy (one HIP runtime per
# process), so torch tensors / streams / RCCL interoperate with these kernels.
import torch  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
# (PLVS_HIP_LIB: a developer build of the same library, e.g. one with kernel phase clocks)
LIB_PATH = os.environ.get("PLVS_HIP_LIB") or os.path.join(_HERE, "lib", "libplvs_hip.so")

PLVS_OK = 0
PLVS_ERR_INVALID_ARG = -1
PLVS_ERR_HIP = -2
PLVS_ERR_NO_DEVICE = -3
PLVS_ERR_CAPACITY = -4
PLVS_ERR_EMPTY = -5
PLVS_ERR_HALO = -6
PLVS_ERR_COMM_FATAL = -7   # a sharded step failed past its counts exchange: abort the communicator on every rank, rebuild the map

TIE_LOWEST_INDEX = 0
TIE_MIH = 1


class PlvsHipError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"plvs_hip error {code}: {msg}")
        self.code = code


class TsdfChiselParams(ctypes.Structure):
    _fields_ = [
        ("resolution", ctypes.c_float),
        ("trunc_quad", ctypes.c_float),
        ("trunc_linear", ctypes.c_float),
        ("trunc_const", ctypes.c_float),
        ("trunc_scale", ctypes.c_float),
        ("weight", ctypes.c_float),
        ("max_chunks", ctypes.c_int32),
        ("shard_rank", ctypes.c_int32),
        ("shard_count", ctypes.c_int32),
        ("order_free", ctypes.c_int32),
    ]


class TsdfStats(ctypes.Structure):
    _fields_ = [
        ("visits", ctypes.c_int64),
        ("points", ctypes.c_int64),
        ("new_chunks", ctypes.c_int32),
        ("updated_chunks", ctypes.c_int32),
        ("voxels", ctypes.c_int32),
        ("max_run", ctypes.c_int32),
    ]


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: build it with `make -C plvs_amd/csrc` "
            "(or __graft_entry__.build()); there is no CPU fallback"
        )
    return ctypes.CDLL(LIB_PATH)


lib = _load()

_vp = ctypes.c_void_p
_i = ctypes.c_int
_ip = ctypes.POINTER(ctypes.c_int)

lib.plvs_hip_last_error.restype = ctypes.c_char_p
lib.plvs_hip_last_error.argtypes = []
lib.plvs_hip_abi_version.restype = _i
lib.plvs_hip_device_count.argtypes = [_ip]
lib.plvs_hip_set_device.argtypes = [_i]
lib.plvs_hip_malloc.argtypes = [ctypes.POINTER(_vp), ctypes.c_size_t]
lib.plvs_hip_free.argtypes = [_vp]
lib.plvs_hip_memcpy_h2d.argtypes = [_vp, _vp, ctypes.c_size_t]
lib.plvs_hip_memcpy_d2h.argtypes = [_vp, _vp, ctypes.c_size_t]
lib.plvs_hip_memset.argtypes = [_vp, _i, ctypes.c_size_t]

lib.plvs_hip_hamming_knn2.argtypes = [_vp, _i, _vp, _i, _vp, _i, _vp, _vp]
lib.plvs_hip_hamming_knn2_dev.argtypes = [_vp, _i, _vp, _i, _vp, _i, _vp, _vp, _vp]

lib.plvs_hip_tsdf_chisel_default_params.argtypes = [ctypes.c_float, ctypes.POINTER(TsdfChiselParams)]
lib.plvs_hip_tsdf_chisel_create.argtypes = [ctypes.POINTER(TsdfChiselParams), ctypes.POINTER(_vp)]
lib.plvs_hip_tsdf_chisel_destroy.argtypes = [_vp]
lib.plvs_hip_tsdf_chisel_clear.argtypes = [_vp]
lib.plvs_hip_tsdf_chisel_integrate.argtypes = [_vp, _vp, _vp, _vp, _i, _vp]
lib.plvs_hip_tsdf_chisel_integrate_batch_dev.argtypes = [_vp, _vp, _vp, _vp, _vp, _i, _vp, _vp]
lib.plvs_hip_tsdf_chisel_last_stats.argtypes = [_vp, ctypes.POINTER(TsdfStats)]
lib.plvs_hip_tsdf_chisel_set_profiling.argtypes = [_vp, _i]
lib.plvs_hip_tsdf_chisel_stage_ms.argtypes = [_vp, _vp, _i, _ip, ctypes.POINTER(ctypes.c_int64)]
lib.plvs_hip_tsdf_chisel_stage_name.argtypes = [_i]
lib.plvs_hip_tsdf_chisel_stage_name.restype = ctypes.c_char_p
lib.plvs_hip_tsdf_chisel_stage_name_of.argtypes = [_vp, _i]
lib.plvs_hip_tsdf_chisel_stage_name_of.restype = ctypes.c_char_p
lib.plvs_hip_tsdf_chisel_updated_chunk_ids_dev.argtypes = [_vp, _vp, _i, _ip, _vp]
lib.plvs_hip_tsdf_chisel_num_chunks.argtypes = [_vp, _ip]
lib.plvs_hip_tsdf_chisel_chunk_ids.argtypes = [_vp, _vp, _i, _ip]
lib.plvs_hip_tsdf_chisel_updated_chunk_ids.argtypes = [_vp, _vp, _i, _ip]
lib.plvs_hip_tsdf_chisel_download_chunk.argtypes = [_vp, _i, _i, _i, _vp, _vp, _vp, _vp]


def check(rc):
    if rc != PLVS_OK:
        raise PlvsHipError(rc, lib.plvs_hip_last_error().decode("utf-8", "replace"))


def np_ptr(a):
    """void* of a C-contiguous numpy array (or None)."""
    if a is None:
        return None
    assert a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(ctypes.c_void_p)


def t_ptr(t):
    """void* of a contiguous torch device tensor (or None)."""
    if t is None:
        return None
    assert t.is_contiguous()
    return ctypes.c_void_p(t.data_ptr())


def current_stream_ptr():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
