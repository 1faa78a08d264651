"""Thin ctypes host layer over liblbm_hip.so (C-ABI: include/lbm_hip.h).

Replaces the reference's pycuda layer: /root/reference/lb_cuda_kernel.py:1-23 and
latticeboltzmann.py:125-159,186-206 (SourceModule JIT, get_function/prepare, drv.to_device,
prepared_async_call, the Python-side buffer swap).  Names follow the reference:
`fused_collide_stream`, `fused_collide_stream_display`, cells / newcells / blocked / surroundings.

There is no fallback: if the shared library is missing, or there is no HIP device, this raises.
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("LBM_HIP_LIB", os.path.join(_HERE, "liblbm_hip.so"))  # env: dev experiments

ABI_VERSION = 3  # include/lbm_hip.h LBM_ABI_VERSION
BOUNDARY_REFERENCE, BOUNDARY_PERIODIC, BOUNDARY_CUDA_KERNEL = 0, 1, 2
YEDGE_DOMAIN, YEDGE_SEAM = 0, 1
STORE_F32, STORE_F16 = 0, 1
LATTICE_D2Q9, LATTICE_D2Q21 = 0, 1
_LATTICES = {"d2q9": (LATTICE_D2Q9, 9), "d2q21": (LATTICE_D2Q21, 21)}
_BOUNDARIES = {"reference": BOUNDARY_REFERENCE, "periodic": BOUNDARY_PERIODIC,
               "cuda_kernel": BOUNDARY_CUDA_KERNEL}
_STORAGE = {np.dtype(np.float32): STORE_F32, np.dtype(np.float16): STORE_F16}

# every symbol include/lbm_hip.h declares (tests check that the library exports all of them)
SYMBOLS = [
    "lbm_abi_version", "lbm_last_error", "lbm_device_count", "lbm_create", "lbm_destroy",
    "lbm_set_stream", "lbm_set_surroundings", "lbm_upload_blocked", "lbm_upload_cells",
    "lbm_init_equilibrium", "lbm_init_blocked_discs", "lbm_fused_collide_stream",
    "lbm_fused_collide_stream_display", "lbm_step_rows", "lbm_fused_collide_stream_timed",
    "lbm_sync", "lbm_steps_done", "lbm_read_cells", "lbm_read_moments", "lbm_read_blocked",
    "lbm_render", "lbm_halo_region", "lbm_mask_halo_region", "lbm_halo_valid",
    "lbm_halo_mark_filled", "lbm_set_tuning", "lbm_advance", "lbm_host_alloc",
    "lbm_host_free", "lbm_advance_rows", "lbm_fused_collide_stream_display_noalloc",
]
NOALLOC_FRAME_MAX_BYTES = 32 << 20  # include/lbm_hip.h LBM_NOALLOC_FRAME_MAX_BYTES


def pagelocked_empty(shape, dtype=np.uint8):
    """drv.pagelocked_empty of latticeboltzmann.py:158-159: a numpy array in page-locked host
    memory (asynchronous device-to-host copies); release_pagelocked frees it."""
    dtype = np.dtype(dtype)
    nbytes = int(np.prod(shape)) * dtype.itemsize
    p = ctypes.c_void_p()
    _check(load_library().lbm_host_alloc(nbytes, ctypes.byref(p)))
    buf = (ctypes.c_ubyte * nbytes).from_address(p.value)
    a = np.frombuffer(buf, dtype=dtype).reshape(shape)
    _PINNED[a.ctypes.data] = p.value
    return a


def release_pagelocked(a):
    p = _PINNED.pop(a.ctypes.data, None)
    if p is not None:
        _check(load_library().lbm_host_free(p))


_PINNED = {}


class lbm_config(ctypes.Structure):
    _fields_ = [
        ("struct_size", ctypes.c_uint32), ("device", ctypes.c_int32), ("N", ctypes.c_int32),
        ("M", ctypes.c_int32), ("omega", ctypes.c_double), ("inner_timesteps", ctypes.c_int32),
        ("boundary", ctypes.c_int32), ("storage", ctypes.c_int32), ("halo", ctypes.c_int32),
        ("top_edge", ctypes.c_int32), ("bottom_edge", ctypes.c_int32),
        ("abi_version", ctypes.c_int32), ("lattice", ctypes.c_int32), ("reserved", ctypes.c_int32 * 2),
    ]


_lib = None


def load_library(path=LIB_PATH):
    """dlopen the HIP library and declare its prototypes.  Fails loudly when it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(path):
        raise RuntimeError(
            f"{path} not found: build it with `make -C latticeboltzmann_amd/csrc` "
            "(or __graft_entry__.build()); there is no CPU fallback")
    # torch wheels bundle their own ROCm runtime; two HIP runtimes in one process cannot both own
    # the GPU.  If torch is going to be used at all (streams, torch.distributed) it has to be
    # loaded first, so that this library binds to the runtime that is already there.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = ctypes.CDLL(path)
    c_int, c_void_p, c_float = ctypes.c_int, ctypes.c_void_p, ctypes.c_float
    fp = ctypes.POINTER(ctypes.c_float)
    bp = ctypes.POINTER(ctypes.c_uint8)
    H = c_void_p
    protos = {
        "lbm_abi_version": ([], c_int),
        "lbm_last_error": ([], ctypes.c_char_p),
        "lbm_device_count": ([ctypes.POINTER(c_int)], c_int),
        "lbm_create": ([ctypes.POINTER(lbm_config), ctypes.POINTER(H)], c_int),
        "lbm_destroy": ([H], c_int),
        "lbm_set_stream": ([H, c_void_p], c_int),
        "lbm_set_surroundings": ([H, fp], c_int),
        "lbm_upload_blocked": ([H, bp, c_int, c_int], c_int),
        "lbm_upload_cells": ([H, fp, c_int, c_int], c_int),
        "lbm_init_equilibrium": ([H, c_float, c_float, c_float, c_float, ctypes.c_uint64,
                                  ctypes.c_int64, ctypes.c_int64], c_int),
        "lbm_init_blocked_discs": ([H, c_int, c_int, ctypes.c_uint64, ctypes.c_int64,
                                    ctypes.c_int64], c_int),
        "lbm_fused_collide_stream": ([H, c_int], c_int),
        "lbm_fused_collide_stream_display": ([H, bp], c_int),
        "lbm_step_rows": ([H, c_int, c_int, c_int], c_int),
        "lbm_fused_collide_stream_timed": ([H, c_int, fp], c_int),
        "lbm_sync": ([H], c_int),
        "lbm_steps_done": ([H, ctypes.POINTER(ctypes.c_int64)], c_int),
        "lbm_read_cells": ([H, fp, c_int, c_int, c_int, c_int], c_int),
        "lbm_read_moments": ([H, fp, fp, c_int, c_int, c_int, c_int], c_int),
        "lbm_read_blocked": ([H, bp, c_int, c_int], c_int),
        "lbm_render": ([H, bp, c_int, c_int], c_int),
        "lbm_halo_region": ([H, c_int, c_int, c_int, ctypes.POINTER(c_void_p),
                             ctypes.POINTER(ctypes.c_size_t)], c_int),
        "lbm_mask_halo_region": ([H, c_int, c_int, ctypes.POINTER(c_void_p),
                                  ctypes.POINTER(ctypes.c_size_t)], c_int),
        "lbm_halo_valid": ([H, ctypes.POINTER(c_int)], c_int),
        "lbm_halo_mark_filled": ([H], c_int),
        "lbm_set_tuning": ([H, c_int, c_int], c_int),
        "lbm_advance": ([H, c_int], c_int),
        "lbm_host_alloc": ([ctypes.c_size_t, ctypes.POINTER(ctypes.c_void_p)], c_int),
        "lbm_host_free": ([ctypes.c_void_p], c_int),
        "lbm_advance_rows": ([H, c_int, c_int, c_int, c_int], c_int),
        "lbm_fused_collide_stream_display_noalloc": ([H, bp], c_int),
    }
    assert set(protos) == set(SYMBOLS)
    for name, (argtypes, restype) in protos.items():
        fn = getattr(L, name)  # AttributeError if the library does not export it
        fn.argtypes = argtypes
        fn.restype = restype
    if L.lbm_abi_version() != ABI_VERSION:
        raise RuntimeError(f"liblbm_hip.so ABI version {L.lbm_abi_version()} != {ABI_VERSION}")
    _lib = L
    return L


def _check(rc):
    """Nonzero status -> RuntimeError, as pycuda raises on launch/driver errors."""
    if rc != 0:
        msg = _lib.lbm_last_error().decode(errors="replace")
        raise RuntimeError(f"liblbm_hip error {rc}: {msg}")


def device_count():
    L = load_library()
    n = ctypes.c_int(0)
    _check(L.lbm_device_count(ctypes.byref(n)))
    return n.value


def _fptr(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_float)) if a is not None else None


def _bptr(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)) if a is not None else None


class Solver:
    """One D2Q9 domain (or one y-slab of it) on one GPU.

    Mirrors what latticeboltzmann.py holds as module globals after :125-159: the compiled kernel
    entry points, cells_gpu / newcells_gpu / blocked_gpu / surroundings_gpu and a stream.
    """

    def __init__(self, N, M, omega, inner_timesteps=1, boundary="reference", dtype=np.float32,
                 device=0, halo=0, top_edge=YEDGE_DOMAIN, bottom_edge=YEDGE_DOMAIN, lattice="d2q9"):
        """lattice="d2q21": the draft lattice of latticeboltzmann.jl:3-22 (periodic, float32, one
        update per launch; cells are (N, M, 21); parity-unpinned, see include/lbm_hip.h)."""
        self._h = None
        self._L = load_library()
        cfg = lbm_config()
        cfg.struct_size = ctypes.sizeof(lbm_config)
        cfg.abi_version = ABI_VERSION
        cfg.device = device
        cfg.N, cfg.M = int(N), int(M)
        cfg.omega = float(omega)
        cfg.inner_timesteps = int(inner_timesteps)
        cfg.boundary = _BOUNDARIES[boundary] if isinstance(boundary, str) else int(boundary)
        cfg.storage = _STORAGE[np.dtype(dtype)]
        cfg.halo = int(halo)
        cfg.top_edge, cfg.bottom_edge = int(top_edge), int(bottom_edge)
        cfg.lattice, self.Q = _LATTICES[lattice]
        h = ctypes.c_void_p()
        _check(self._L.lbm_create(ctypes.byref(cfg), ctypes.byref(h)))
        self._h = h
        self.N, self.M, self.halo = int(N), int(M), int(halo)
        self.inner_timesteps = int(inner_timesteps)
        self.device = device

    # -- lifetime ----------------------------------------------------------------------------
    def close(self):
        if self._h is not None:
            self._L.lbm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def set_stream(self, hip_stream):
        """hip_stream: integer hipStream_t (e.g. torch.cuda.Stream().cuda_stream) or None."""
        _check(self._L.lbm_set_stream(self._h, ctypes.c_void_p(hip_stream) if hip_stream else None))

    # -- upload (drv.to_device, latticeboltzmann.py:147-150) --------------------------------------
    def set_surroundings(self, surroundings):
        s = np.ascontiguousarray(surroundings, dtype=np.float32).reshape(self.Q)
        _check(self._L.lbm_set_surroundings(self._h, _fptr(s)))

    def upload_blocked(self, blocked, row0=0):
        b = np.ascontiguousarray(blocked, dtype=np.uint8)
        assert b.ndim == 2 and b.shape[1] == self.M, b.shape
        _check(self._L.lbm_upload_blocked(self._h, _bptr(b), row0, b.shape[0]))

    def upload_cells(self, cells, row0=0):
        c = np.ascontiguousarray(cells, dtype=np.float32)
        assert c.ndim == 3 and c.shape[1] == self.M and c.shape[2] == self.Q, c.shape
        _check(self._L.lbm_upload_cells(self._h, _fptr(c), row0, c.shape[0]))

    def init_equilibrium(self, rho, u, amp=0.0, seed=0, row_offset=0, global_rows=0):
        _check(self._L.lbm_init_equilibrium(self._h, rho, u[0], u[1], amp, seed, row_offset,
                                            global_rows))

    def init_blocked_discs(self, ndiscs, radius, seed=0, row_offset=0, global_rows=0):
        _check(self._L.lbm_init_blocked_discs(self._h, ndiscs, radius, seed, row_offset, global_rows))

    # -- the hot path (prepared_async_call + swap, latticeboltzmann.py:187-202) -------------------
    def fused_collide_stream(self, n_outer=1):
        _check(self._L.lbm_fused_collide_stream(self._h, n_outer))

    def advance(self, n_updates):
        """Exactly n_updates lattice updates (deepest launches the tuning / ghost rows allow)."""
        _check(self._L.lbm_advance(self._h, n_updates))

    def fused_collide_stream_display(self, frame=None):
        """frame: None (plain step, latticeboltzmann.py:195-196) or a C-contiguous uint8 (N, M, 3)
        host array filled asynchronously (valid after sync())."""
        if frame is not None:
            assert frame.dtype == np.uint8 and frame.flags.c_contiguous
            assert frame.shape == (self.N, self.M, 3), frame.shape
        _check(self._L.lbm_fused_collide_stream_display(self._h, _bptr(frame)))

    def fused_collide_stream_display_noalloc(self, frame=None, device_ptr=None):
        """The same without any allocation in the library (static device scratch): `frame` must be
        page-locked (pagelocked_empty) and at most NOALLOC_FRAME_MAX_BYTES.  device_ptr: the address
        of N*M*3 bytes of DEVICE memory to receive the frame instead (e.g. tensor.data_ptr())."""
        if device_ptr is not None:
            assert frame is None
            p = ctypes.cast(ctypes.c_void_p(int(device_ptr)), ctypes.POINTER(ctypes.c_uint8))
            _check(self._L.lbm_fused_collide_stream_display_noalloc(self._h, p))
            return
        if frame is not None:
            assert frame.dtype == np.uint8 and frame.flags.c_contiguous
            assert frame.shape == (self.N, self.M, 3), frame.shape
        _check(self._L.lbm_fused_collide_stream_display_noalloc(self._h, _bptr(frame)))

    def set_tuning(self, tb_max_depth=0, tb_rows=0):
        _check(self._L.lbm_set_tuning(self._h, tb_max_depth, tb_rows))

    def step_rows(self, row0, nrows, finish):
        """One update of rows [row0, row0+nrows) into the other buffer; finish=True closes it."""
        _check(self._L.lbm_step_rows(self._h, row0, nrows, int(bool(finish))))

    def advance_rows(self, row0, nrows, depth, finish):
        """`depth` updates of rows [row0, row0+nrows) in one launch into the other buffer."""
        _check(self._L.lbm_advance_rows(self._h, row0, nrows, depth, int(bool(finish))))

    def fused_collide_stream_timed(self, n_outer):
        """Returns the device time in ms (HIP events on the solver's stream)."""
        ms = ctypes.c_float(0)
        _check(self._L.lbm_fused_collide_stream_timed(self._h, n_outer, ctypes.byref(ms)))
        return ms.value

    def sync(self):
        _check(self._L.lbm_sync(self._h))

    @property
    def steps_done(self):
        n = ctypes.c_int64(0)
        _check(self._L.lbm_steps_done(self._h, ctypes.byref(n)))
        return n.value

    # -- readback ----------------------------------------------------------------------------------
    def _window(self, rows, cols):
        r0, r1 = rows if rows is not None else (0, self.N)
        c0, c1 = cols if cols is not None else (0, self.M)
        return r0, r1 - r0, c0, c1 - c0

    def read_cells(self, rows=None, cols=None):
        r0, nr, c0, nc = self._window(rows, cols)
        out = np.empty((nr, nc, self.Q), np.float32)
        _check(self._L.lbm_read_cells(self._h, _fptr(out), r0, nr, c0, nc))
        return out

    def read_moments(self, rows=None, cols=None):
        """Returns (u (nr, nc, 2) [row, col], rho (nr, nc)) like latticeboltzmann.py:180-183."""
        r0, nr, c0, nc = self._window(rows, cols)
        rho = np.empty((nr, nc), np.float32)
        u = np.empty((nr, nc, 2), np.float32)
        _check(self._L.lbm_read_moments(self._h, _fptr(rho), _fptr(u), r0, nr, c0, nc))
        return u, rho

    def read_blocked(self, row0=0, nrows=None):
        nrows = self.N - row0 if nrows is None else nrows
        out = np.empty((nrows, self.M), np.uint8)
        _check(self._L.lbm_read_blocked(self._h, _bptr(out), row0, nrows))
        return out.astype(bool)

    def render(self, row0=0, nrows=None):
        nrows = self.N - row0 if nrows is None else nrows
        out = np.empty((nrows, self.M, 3), np.uint8)
        _check(self._L.lbm_render(self._h, _bptr(out), row0, nrows))
        return out

    # -- y-slab halos ------------------------------------------------------------------------------
    def halo_region(self, side, recv, next=False):
        """(device pointer, bytes) of the send (recv=False) or ghost (recv=True) rows at
        side 0 (top) / 1 (bottom) of the current state, or of the buffer an open update is
        writing (next=True)."""
        p = ctypes.c_void_p()
        n = ctypes.c_size_t()
        _check(self._L.lbm_halo_region(self._h, side, int(recv), int(next), ctypes.byref(p),
                                       ctypes.byref(n)))
        return p.value, n.value

    def mask_halo_region(self, side, recv):
        p = ctypes.c_void_p()
        n = ctypes.c_size_t()
        _check(self._L.lbm_mask_halo_region(self._h, side, int(recv), ctypes.byref(p),
                                            ctypes.byref(n)))
        return p.value, n.value

    @property
    def halo_valid(self):
        n = ctypes.c_int(0)
        _check(self._L.lbm_halo_valid(self._h, ctypes.byref(n)))
        return n.value

    def halo_mark_filled(self):
        _check(self._L.lbm_halo_mark_filled(self._h))
