"""MI355X-native spectral-integration hot path — Python host mirror over the C ABI.

The product is ``libsri_hip.so`` (``csrc/``: C++ host formers + the gfx950 MFMA-fp64 apply kernel,
C ABI declared in ``include/sri_capi.h``).  This package is a thin ctypes binding of that ABI with
the reference's own names (``ComputeChebyshevPoints``, ``GetCoefficients_c``, ``getDn`` —
``/root/reference/include/chebyshev_differentiation.h:19-108``) plus the batched apply that
replaces ``main.cpp:172`` / ``main.cpp:97``.  PyTorch is used only for device memory and streams.

There is NO CPU fallback for the apply: if the shared library is missing, or no gfx950 device is
present, the calls raise.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Tuple

import numpy as np

__all__ = [
    "SriError", "build_library", "library_path", "load_library",
    "ComputeChebyshevPoints", "GetCoefficients_c", "getDn", "dn_blocks", "integration_matrix", "ivp",
    "SpectralApply", "RodIntegrator", "OP_DN", "OP_DNN_INV", "OP_DNN",
    "OPT_COLUMN_PLAN", "OPT_LEAN_TAIL", "OPT_ROW_PARTS", "OPT_SMALL_KERNEL", "OPT_XCD_TILE_WALK", "OPT_HOST_COPY_THREADS",
    "OPT_HOST_BLOCK_COLUMNS", "OPT_MAX_WORKGROUPS", "OPT_PLAN_CALIBRATION", "ROD_OPT_FIRST_SWEEP_CHECK", "ROD_OPT_FUSED_POSITION_RHS", "ROD_OPT_FUSED_CRITERION",
]

_PKG = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_PKG)
_LIB_NAME = "libsri_hip.so"

OP_DN, OP_DNN_INV, OP_DNN = 0, 1, 2
# sri_option (include/sri_capi.h)
(OPT_COLUMN_PLAN, OPT_LEAN_TAIL, OPT_ROW_PARTS, OPT_SMALL_KERNEL, OPT_XCD_TILE_WALK, OPT_HOST_COPY_THREADS, OPT_HOST_BLOCK_COLUMNS,
 OPT_MAX_WORKGROUPS, OPT_PLAN_CALIBRATION) = range(1, 10)
ROD_OPT_FIRST_SWEEP_CHECK, ROD_OPT_FUSED_POSITION_RHS, ROD_OPT_FUSED_CRITERION = 100, 101, 102   # sri_rod_set_option only

_dp = C.POINTER(C.c_double)
_lib: Optional[C.CDLL] = None

# name -> (restype, argtypes): every symbol include/sri_capi.h declares.
_u32, _sz, _dbl, _int, _vp = C.c_uint32, C.c_size_t, C.c_double, C.c_int, C.c_void_p
C_ABI = {
    "sri_cheb_points": (_int, [_u32, _dbl, _dp]),
    "sri_cheb_coeffs": (_int, [_u32, _dp]),
    "sri_cheb_dn": (_int, [_u32, _dp, _sz]),
    "sri_cheb_blocks": (_int, [_u32, _dp, _sz, _dp, _sz, _dp]),
    "sri_cheb_integration": (_int, [_u32, _dp, _sz, _dp]),
    "sri_ivp": (_int, [_u32, _u32, _dp, _dp, _dp, _sz]),
    "sri_cheb_interp": (_int, [_u32, _u32, _dp, _sz]),
    "sri_cheb_coarse_view": (_int, [_u32, _dp, _sz, _dp, _u32, _dp, _sz, _dp]),
    "sri_apply_create": (_int, [C.POINTER(_vp), _u32, _int, _int]),
    "sri_apply_create_matrix": (_int, [C.POINTER(_vp), _u32, _dp, _sz, _int]),
    "sri_apply_set_max_workgroups": (_int, [_vp, _u32]),
    "sri_apply_dim": (_int, [_vp, C.POINTER(_u32)]),
    "sri_apply": (_int, [_vp, _vp, _sz, _vp, _sz, _sz, _vp]),
    "sri_apply_tips": (_int, [_vp, _vp, _sz, _vp, _sz, _sz, _vp, _vp]),
    "sri_apply_many": (_int, [_vp, _vp, _sz, _vp, _sz, _sz, _sz, _vp]),
    "sri_integrate": (_int, [_vp, _vp, _sz, _vp, _vp, _sz, _sz, _vp]),
    "sri_integrate_tips": (_int, [_vp, _vp, _sz, _vp, _vp, _sz, _sz, _vp, _vp]),
    "sri_apply_host": (_int, [_vp, _dp, _sz, _dp, _sz, _sz]),
    "sri_integrate_host": (_int, [_vp, _dp, _sz, _dp, _dp, _sz, _sz]),
    "sri_apply_set_lean_tail": (_int, [_vp, _int]),
    "sri_apply_set_column_plan": (_int, [_vp, _int]),
    "sri_apply_set_host_copy_threads": (_int, [_vp, _int]),
    "sri_apply_set_host_memory_pinned": (_int, [_vp, _int]),
    "sri_apply_host_trim": (_int, [_vp]),
    "sri_apply_set_option": (_int, [_vp, _int, C.c_long]),
    "sri_apply_get_option": (_int, [_vp, _int, C.POINTER(C.c_long)]),
    "sri_rod_set_option": (_int, [_vp, _int, C.c_long]),
    "sri_rod_get_option": (_int, [_vp, _int, C.POINTER(C.c_long)]),
    "sri_apply_destroy": (_int, [_vp]),
    "sri_rod_create": (_int, [C.POINTER(_vp), _u32, _int]),
    "sri_rod_destroy": (_int, [_vp]),
    "sri_rod_strain_field": (_int, [_vp, _vp, _sz, _vp, _sz, _vp]),
    "sri_rod_quaternion_rhs": (_int, [_vp, _vp, _sz, _vp, _sz, _sz, _vp, _sz, _vp]),
    "sri_rod_set_resident": (_int, [_vp, _int]),
    "sri_rod_set_predictor": (_int, [_vp, _int]),
    "sri_rod_last_apply_kernel": (C.c_char_p, [_vp]),
    "sri_rod_quaternion_step": (_int, [_vp, _vp, _vp, _sz, _vp, _sz, _vp, _sz, _vp, _vp, _vp, _vp]),
    "sri_rod_position_rhs": (_int, [_vp, _vp, _sz, _sz, _vp, _sz, _vp]),
    "sri_rod_integrate_quaternions": (_int, [_vp, _vp, _sz, _dp, _vp, _sz, _dbl, _u32, C.POINTER(_u32), _dp, _vp]),
    "sri_rod_integrate_positions": (_int, [_vp, _vp, _sz, _sz, _dp, _vp, _sz, _vp]),
    "sri_rod_integrate": (_int, [_vp, _vp, _sz, _dp, _dp, _vp, _sz, _vp, _sz, _dbl, _u32, C.POINTER(_u32), _dp, _vp]),
    "sri_comm_unique_id": (_int, [_vp]),
    "sri_comm_create": (_int, [C.POINTER(_vp), _vp, _int, _int, _int]),
    "sri_comm_destroy": (_int, [_vp]),
    "sri_comm_rank": (_int, [_vp, C.POINTER(_int), C.POINTER(_int)]),
    "sri_allgather_tips": (_int, [_vp, _vp, _sz, _vp, C.POINTER(_sz), _vp]),
    "sri_allgather_panel": (_int, [_vp, _vp, _sz, _sz, _vp, C.POINTER(_sz), _vp]),
    "sri_last_rccl_error": (_int, []),
    "sri_rccl_strerror": (C.c_char_p, [_int]),
    "sri_apply_last_kernel": (C.c_char_p, [_vp]),
    "sri_apply_summation_order": (_int, [_vp, C.POINTER(_int)]),
    "sri_strerror": (C.c_char_p, [_int]),
    "sri_last_hip_error": (_int, []),
    "sri_abi_version": (_int, []),
    "sri_device_count": (_int, [C.POINTER(_int)]),
}


class SriError(RuntimeError):
    def __init__(self, status: int, what: str):
        self.status = status
        msg = "?"
        if _lib is not None:
            msg = _lib.sri_strerror(status).decode()
            if status == -4:
                msg += f" (hipError {_lib.sri_last_hip_error()})"
            if status == -8:
                e = _lib.sri_last_rccl_error()
                msg += f" (ncclResult {e}: {_lib.sri_rccl_strerror(e).decode()})"
        super().__init__(f"{what}: {msg} [sri_status {status}]")


def library_path() -> str:
    # SRI_HIP_LIBRARY: an alternative build of the SAME library (A/B measurements of kernel variants, tools/ only)
    return os.environ.get("SRI_HIP_LIBRARY") or os.path.join(_PKG, _LIB_NAME)


def build_library(force: bool = False) -> str:
    """Compile ``libsri_hip.so`` for gfx950 with hipcc (cross-compiles without a GPU)."""
    import subprocess

    csrc = os.path.join(_PKG, "csrc")
    cmd = ["make", "-j4", "-C", csrc] + (["-B"] if force else [])
    subprocess.check_call(cmd, stdout=subprocess.DEVNULL)
    return library_path()


def load_library() -> C.CDLL:
    """dlopen the product library; raises (never falls back) if it has not been built."""
    global _lib
    if _lib is None:
        path = library_path()
        if not os.path.exists(path):
            raise ImportError(
                f"{path} is missing: the HIP extension has not been built "
                f"(run `python -c 'import __graft_entry__ as g; g.build()'` or `make -C {os.path.join(_PKG, 'csrc')}`). "
                "There is no CPU fallback for the spectral apply.")
        # One HIP runtime per process: PyTorch-ROCm wheels bundle their own libamdhip64.so.7 /
        # libhsa-runtime64.so.1.  Import torch FIRST so that our DT_NEEDED entries resolve (by
        # soname) to the copies torch already mapped; loading /opt/rocm's copies next to torch's
        # gives two runtimes fighting over the device ("No HIP GPUs are available").
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        lib = C.CDLL(path)
        for name, (res, args) in C_ABI.items():
            fn = getattr(lib, name)   # AttributeError if the library does not export the symbol
            fn.restype, fn.argtypes = res, args
        _lib = lib
    return _lib


def _chk(status: int, what: str) -> None:
    if status != 0:
        raise SriError(status, what)


def _fmat(rows: int, cols: int) -> np.ndarray:
    return np.zeros((rows, cols), dtype=np.float64, order="F")


def _p(a: np.ndarray):
    return a.ctypes.data_as(_dp)


# ------------------------------------------------------------------------------------------------
# Host formers — same names as the reference header (chebyshev_differentiation.h)
# ------------------------------------------------------------------------------------------------
def ComputeChebyshevPoints(n: int, L: float = 1.0) -> np.ndarray:
    """``ComputeChebyshevPoints<N, L>()`` (chebyshev_differentiation.h:19-30)."""
    x = np.zeros(int(n), dtype=np.float64)
    _chk(load_library().sri_cheb_points(n, L, _p(x)), "sri_cheb_points")
    return x


def GetCoefficients_c(n: int) -> np.ndarray:
    """``GetCoefficients_c<N>()`` (chebyshev_differentiation.h:37-52)."""
    c = np.zeros(int(n), dtype=np.float64)
    _chk(load_library().sri_cheb_coeffs(n, _p(c)), "sri_cheb_coeffs")
    return c


def getDn(n: int) -> np.ndarray:
    """``getDn<N>()`` (chebyshev_differentiation.h:59-108): column-major N x N ndarray."""
    D = _fmat(n, n)
    _chk(load_library().sri_cheb_dn(n, _p(D), n), "sri_cheb_dn")
    return D


def dn_blocks(D: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """``Dn_NN``, ``Dn_IN`` of main.cpp:94-95."""
    D = np.asfortranarray(D, dtype=np.float64)
    n = D.shape[0]
    Dnn, Din = _fmat(n - 1, n - 1), np.zeros(n - 1)
    _chk(load_library().sri_cheb_blocks(n, _p(D), n, _p(Dnn), n - 1, _p(Din)), "sri_cheb_blocks")
    return Dnn, Din


def integration_matrix(n: int) -> Tuple[np.ndarray, np.ndarray]:
    """``(Dn_NN.inverse(), Dn_IN)`` of main.cpp:157-160."""
    Dinv, Din = _fmat(n - 1, n - 1), np.zeros(n - 1)
    _chk(load_library().sri_cheb_integration(n, _p(Dinv), n - 1, _p(Din)), "sri_cheb_integration")
    return Dinv, Din


def ivp(Din: np.ndarray, x_base: np.ndarray) -> np.ndarray:
    """Initial-value term ``Dn_IN * x_base^T`` (main.cpp:109, 162-164)."""
    Din = np.ascontiguousarray(Din, dtype=np.float64)
    x_base = np.ascontiguousarray(x_base, dtype=np.float64)
    out = _fmat(Din.shape[0], x_base.shape[0])
    _chk(load_library().sri_ivp(Din.shape[0], x_base.shape[0], _p(Din), _p(x_base), _p(out), Din.shape[0]), "sri_ivp")
    return out


def cheb_interp(n_from: int, n_to: int) -> np.ndarray:
    """Barycentric interpolation matrix (n_to x n_from) between two Chebyshev grids of the reference on [0, 1]."""
    P = _fmat(n_to, n_from)
    _chk(load_library().sri_cheb_interp(n_from, n_to, _p(P), n_to), "sri_cheb_interp")
    return P


def cheb_coarse_view(A_fine: np.ndarray, w_fine: np.ndarray, n_coarse: int) -> Tuple[np.ndarray, np.ndarray]:
    """(A_coarse, w_coarse): the integration operator ``A_fine`` ((n_fine-1)^2) and its initial-value weights as a grid of
    ``n_coarse`` nodes sees them (sri_cheb_coarse_view) — what the rod predictor's coarse solve runs on."""
    A_fine = np.asfortranarray(A_fine, dtype=np.float64)
    w_fine = np.ascontiguousarray(w_fine, dtype=np.float64)
    n_fine = A_fine.shape[0] + 1
    if A_fine.shape != (n_fine - 1, n_fine - 1) or w_fine.shape != (n_fine - 1,):
        raise ValueError("A_fine must be (n_fine-1, n_fine-1) and w_fine (n_fine-1,)")
    Ac, wc = _fmat(n_coarse - 1, n_coarse - 1), np.zeros(n_coarse - 1)
    _chk(load_library().sri_cheb_coarse_view(n_fine, _p(A_fine), n_fine - 1, _p(w_fine), n_coarse, _p(Ac), n_coarse - 1, _p(wc)),
         "sri_cheb_coarse_view")
    return Ac, wc


# ------------------------------------------------------------------------------------------------
# The batched apply
# ------------------------------------------------------------------------------------------------
class SpectralApply:
    """Device-resident spectral operator: ``Y = A @ X`` for a panel of rod-state columns.

    Panels are torch CUDA tensors of dtype float64 and shape ``(..., rows)`` whose last dimension
    is contiguous — e.g. ``(B, s, N)``: rod, state component, node.  In memory that IS the
    reference's column-major ``N x (s*B)`` matrix (each state column = N contiguous doubles,
    main.cpp:113,130-133), so no transposition happens anywhere.
    """

    def __init__(self, n: int, which: int = OP_DN, device: int = 0, matrix: Optional[np.ndarray] = None):
        lib = load_library()
        self._h = _vp()
        self.device = int(device)
        self.n = int(n)
        self.which = which
        if matrix is not None:
            A = np.asfortranarray(matrix, dtype=np.float64)
            assert A.shape[0] == A.shape[1]
            _chk(lib.sri_apply_create_matrix(C.byref(self._h), A.shape[0], _p(A), A.shape[0], self.device),
                 "sri_apply_create_matrix")
        else:
            _chk(lib.sri_apply_create(C.byref(self._h), self.n, which, self.device), "sri_apply_create")
        r = _u32()
        _chk(lib.sri_apply_dim(self._h, C.byref(r)), "sri_apply_dim")
        self.rows = int(r.value)

    # -- device path ---------------------------------------------------------------------------
    def _panel(self, t, name):
        import torch

        if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.float64):
            raise TypeError(f"{name} must be a float64 CUDA tensor")
        if t.shape[-1] != self.rows:
            raise ValueError(f"{name}: last dimension {t.shape[-1]} != operator rows {self.rows}")
        if not t.is_contiguous():
            raise ValueError(f"{name} must be contiguous (columns of {self.rows} doubles)")
        if t.device.index != self.device:
            raise ValueError(f"{name} is on cuda:{t.device.index}, operator on cuda:{self.device}")
        return t

    def apply(self, X, out=None, stream=None, tips=None):
        """``Y = A @ X`` on the GPU (enqueued on torch's current stream unless ``stream`` is given).

        ``tips``: optional float64 CUDA tensor with one entry per column; receives ``Y[..., 0]``, the
        tip-node value of every column, written by the kernel's epilogue.
        """
        import torch

        X = self._panel(X, "X")
        Y = torch.empty_like(X) if out is None else self._panel(out, "out")
        if Y.shape != X.shape:
            raise ValueError("out must have X's shape")
        M = X.numel() // self.rows
        s = torch.cuda.current_stream(X.device).cuda_stream if stream is None else stream
        if tips is None:
            _chk(load_library().sri_apply(self._h, X.data_ptr(), self.rows, Y.data_ptr(), self.rows, M, s), "sri_apply")
        else:
            if not (tips.is_cuda and tips.dtype == torch.float64 and tips.is_contiguous() and tips.numel() == M):
                raise ValueError("tips must be a contiguous float64 CUDA tensor with one entry per column")
            _chk(load_library().sri_apply_tips(self._h, X.data_ptr(), self.rows, Y.data_ptr(), self.rows, M,
                                               tips.data_ptr(), s), "sri_apply_tips")
        return Y

    __call__ = apply

    def apply_many(self, Xs, outs=None, stream=None):
        """``sri_apply_many``: ``Y_i = A @ X_i`` for a list of equally shaped panels in ONE persistent launch.

        Returns the list of outputs.  The device arrays of panel pointers are built here (two small host-to-device
        copies); a caller that reuses the same panels should keep them and call the C ABI directly."""
        import torch

        Xs = [self._panel(X, "X") for X in Xs]
        if not Xs:
            return []
        if any(X.shape != Xs[0].shape or X.device != Xs[0].device for X in Xs):
            raise ValueError("all panels must have the same shape and device")
        Ys = [torch.empty_like(X) for X in Xs] if outs is None else [self._panel(Y, "out") for Y in outs]
        if len(Ys) != len(Xs) or any(Y.shape != Xs[0].shape for Y in Ys):
            raise ValueError("outs must match Xs")
        dev = Xs[0].device
        px = torch.tensor([X.data_ptr() for X in Xs], dtype=torch.int64).to(dev)
        py = torch.tensor([Y.data_ptr() for Y in Ys], dtype=torch.int64).to(dev)
        M = Xs[0].numel() // self.rows
        s = torch.cuda.current_stream(dev).cuda_stream if stream is None else stream
        _chk(load_library().sri_apply_many(self._h, px.data_ptr(), self.rows, py.data_ptr(), self.rows, M, len(Xs), s), "sri_apply_many")
        self._keep = (px, py)                                  # alive until the next call: the launch reads them asynchronously
        return Ys

    def integrate(self, B, x_base, out=None, stream=None):
        """``Y = Dnn_inv @ (B - Dn_IN x_base^T)`` (main.cpp:172), operator must be ``OP_DNN_INV``.

        ``x_base`` holds the base-node value of every column: shape ``B.shape[:-1]``.
        """
        import torch

        B = self._panel(B, "B")
        if not (x_base.is_cuda and x_base.dtype == torch.float64 and x_base.is_contiguous()
                and tuple(x_base.shape) == tuple(B.shape[:-1])):
            raise ValueError("x_base must be a contiguous float64 CUDA tensor of shape B.shape[:-1]")
        Y = torch.empty_like(B) if out is None else self._panel(out, "out")
        M = B.numel() // self.rows
        s = torch.cuda.current_stream(B.device).cuda_stream if stream is None else stream
        _chk(load_library().sri_integrate(self._h, B.data_ptr(), self.rows, x_base.data_ptr(), Y.data_ptr(),
                                          self.rows, M, s), "sri_integrate")
        return Y

    # -- host convenience (PCIe-inclusive) -------------------------------------------------------
    @staticmethod
    def _host_panel(a, rows: int, what: str):
        """(array, leading dimension, columns) of a column-major host panel: a Fortran-ordered matrix or a column slice of one."""
        if not (isinstance(a, np.ndarray) and a.dtype == np.float64 and a.ndim == 2 and a.shape[0] == rows):
            raise ValueError(f"{what} must be a float64 matrix with {rows} rows")
        if a.size == 0:
            return a, rows, a.shape[1]
        if a.shape[1] > 1 and (a.strides[0] != 8 or a.strides[1] % 8 or a.strides[1] < 8 * rows):
            raise ValueError(f"{what} must be column-major (a Fortran-ordered matrix, or the leading rows of one)")
        if a.shape[1] <= 1 and a.shape[0] > 1 and a.strides[0] != 8:
            raise ValueError(f"{what} must be column-major")
        ld = a.strides[1] // 8 if a.shape[1] > 1 else rows
        return a, ld, a.shape[1]

    def apply_host(self, X: np.ndarray, out: Optional[np.ndarray] = None) -> np.ndarray:
        """``Y = A @ X`` for a column-major host matrix ``rows x M``; X and ``out`` may be the leading rows of a taller
        Fortran-ordered matrix (leading dimension > rows).  The panel crosses PCIe in blocks while the previous block is
        applied and the one before returns (``sri_apply_host``)."""
        if not (isinstance(X, np.ndarray) and X.dtype == np.float64 and X.ndim == 2 and X.strides[0] == 8):
            X = np.asfortranarray(X, dtype=np.float64)
        X, ldx, M = self._host_panel(X, self.rows, "X")
        Y = np.empty((self.rows, M), dtype=np.float64, order="F") if out is None else out
        Y, ldy, My = self._host_panel(Y, self.rows, "out")
        if My != M:
            raise ValueError("out must have as many columns as X")
        _chk(load_library().sri_apply_host(self._h, _p(X), ldx, _p(Y), ldy, M), "sri_apply_host")
        return Y

    def integrate_host(self, B: np.ndarray, x_base: np.ndarray, out: Optional[np.ndarray] = None) -> np.ndarray:
        """Host-panel form of :meth:`integrate` (``sri_integrate_host``): ``Dn_NN⁻¹·B − (Dn_NN⁻¹·Dn_IN)·x_baseᵀ``."""
        if not (isinstance(B, np.ndarray) and B.dtype == np.float64 and B.ndim == 2 and B.strides[0] == 8):
            B = np.asfortranarray(B, dtype=np.float64)
        B, ldb, M = self._host_panel(B, self.rows, "B")
        xb = np.ascontiguousarray(x_base, dtype=np.float64)
        if xb.shape != (M,):
            raise ValueError("x_base must hold one value per column")
        Y = np.empty((self.rows, M), dtype=np.float64, order="F") if out is None else out
        Y, ldy, My = self._host_panel(Y, self.rows, "out")
        if My != M:
            raise ValueError("out must have as many columns as B")
        _chk(load_library().sri_integrate_host(self._h, _p(B), ldb, _p(xb), _p(Y), ldy, M), "sri_integrate_host")
        return Y

    def set_lean_tail(self, enable: bool) -> None:
        """Odd-row-count operators: allow (default) or forbid the kernel variants that read the panel's last column up to its
        last k chunk's end where that cannot fault (``sri_apply_set_lean_tail``); same results either way."""
        _chk(load_library().sri_apply_set_lean_tail(self._h, 1 if enable else 0), "sri_apply_set_lean_tail")

    def set_column_plan(self, enable: bool) -> None:
        """Allow (default) or forbid giving the columns behind the operator's full rounds of tiles — or a small panel as a
        whole — to a finer tile geometry of the same operator (``sri_apply_set_column_plan``); same results either way."""
        _chk(load_library().sri_apply_set_column_plan(self._h, 1 if enable else 0), "sri_apply_set_column_plan")

    def set_host_copy_threads(self, threads: int) -> None:
        """CPU threads that move host panels into / out of the handle's pinned staging slots (-1 = default, 0 = caller only)."""
        _chk(load_library().sri_apply_set_host_copy_threads(self._h, int(threads)), "sri_apply_set_host_copy_threads")

    def host_trim(self) -> None:
        """Give back the pinned staging ring and device slots of the host-panel path (``sri_apply_host_trim``)."""
        _chk(load_library().sri_apply_host_trim(self._h), "sri_apply_host_trim")

    def set_host_memory_pinned(self, pinned: bool) -> None:
        """Promise that every host panel passed to this handle is page-locked (``hipHostMalloc`` / the caller's own
        ``hipHostRegister``): copies then go straight from and to it.  A panel that is not is refused (``SRI_EINVAL``)."""
        _chk(load_library().sri_apply_set_host_memory_pinned(self._h, 1 if pinned else 0), "sri_apply_set_host_memory_pinned")

    def set_option(self, option: int, value: int) -> None:
        """``sri_apply_set_option``: any per-handle switch by its ``OPT_*`` number."""
        _chk(load_library().sri_apply_set_option(self._h, int(option), int(value)), "sri_apply_set_option")

    def get_option(self, option: int) -> int:
        v = C.c_long()
        _chk(load_library().sri_apply_get_option(self._h, int(option), C.byref(v)), "sri_apply_get_option")
        return int(v.value)

    def set_max_workgroups(self, n: int) -> None:
        """Cap the persistent grid (0 = fill the chip): leaves CUs free for an overlapped collective."""
        _chk(load_library().sri_apply_set_max_workgroups(self._h, int(n)), "sri_apply_set_max_workgroups")

    @property
    def last_kernel(self) -> str:
        return load_library().sri_apply_last_kernel(self._h).decode()

    @property
    def ascending_k(self) -> bool:
        """True if launches sum k = 0, 1, 2, ... like main.cpp:172; False: evens before odds inside blocks of 8 k."""
        a = _int()
        _chk(load_library().sri_apply_summation_order(self._h, C.byref(a)), "sri_apply_summation_order")
        return bool(a.value)

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            load_library().sri_apply_destroy(self._h)
            self._h = _vp()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ------------------------------------------------------------------------------------------------
# Batched rod integration: main.cpp's integrateQuaternions + integratePosition
# ------------------------------------------------------------------------------------------------
class RodIntegrator:
    """Integrates a batch of Cosserat rods on the GPU, given their strain coordinates ``qe``.

    ``qe``: float64 CUDA tensor ``(rods, 9)`` in the order of main.cpp:187-195.  Results are panels
    ``Q (rods, 4, N-1)`` (w, x, y, z at the unknown nodes, tip first) and ``r (rods, 3, N-1)`` — the
    column-major ``(N-1) x 4`` / ``(N-1) x 3`` matrices main.cpp prints, one per rod.
    """

    def __init__(self, n: int, device: int = 0):
        self._h = _vp()
        self.n, self.m, self.device = int(n), int(n) - 1, int(device)
        _chk(load_library().sri_rod_create(C.byref(self._h), self.n, self.device), "sri_rod_create")
        self.iterations = 0
        self.last_update = 0.0

    def _stream(self, t):
        import torch
        return torch.cuda.current_stream(t.device).cuda_stream

    def set_resident(self, enable: bool):
        """Small grids (N <= 64) use the on-chip one-launch pipeline; False forces the streaming path."""
        _chk(load_library().sri_rod_set_resident(self._h, 1 if enable else 0), "sri_rod_set_resident")

    def set_predictor(self, enable: bool):
        """Start the fixed point from an interpolated coarse solution (the default where there is one): a 32-node solution for
        N > 64; inside the one-launch pipeline a 32-node one for 32 < N <= 64 and a 16-node one for 16 < N <= 32.  False starts
        from q_init."""
        _chk(load_library().sri_rod_set_predictor(self._h, 1 if enable else 0), "sri_rod_set_predictor")

    def set_option(self, option: int, value: int) -> None:
        """``sri_rod_set_option``: an ``OPT_*`` switch on every apply handle inside this rod handle."""
        _chk(load_library().sri_rod_set_option(self._h, int(option), int(value)), "sri_rod_set_option")

    def get_option(self, option):
        """``sri_rod_get_option``."""
        v = C.c_long(0)
        _chk(load_library().sri_rod_get_option(self._h, int(option), C.byref(v)), "sri_rod_get_option")
        return int(v.value)

    @property
    def last_apply_kernel(self) -> str:
        """What the handle's integration operator launched last (``sri_rod_last_apply_kernel``): a fused step or a plain apply."""
        return load_library().sri_rod_last_apply_kernel(self._h).decode()

    def _check_qe(self, qe):
        import torch
        if not (isinstance(qe, torch.Tensor) and qe.is_cuda and qe.dtype == torch.float64 and qe.is_contiguous()
                and qe.dim() == 2 and qe.shape[1] == 9 and qe.device.index == self.device):
            raise TypeError("qe must be a contiguous float64 CUDA tensor of shape (rods, 9) on the handle's device")
        return qe

    def strain_field(self, qe):
        import torch
        qe = self._check_qe(qe)
        K = torch.empty((qe.shape[0], 3, self.m), dtype=torch.float64, device=qe.device)
        _chk(load_library().sri_rod_strain_field(self._h, qe.data_ptr(), qe.shape[0], K.data_ptr(), self.m, self._stream(qe)),
             "sri_rod_strain_field")
        return K

    def quaternion_rhs(self, K, Q):
        import torch
        G = torch.empty_like(Q)
        _chk(load_library().sri_rod_quaternion_rhs(self._h, K.data_ptr(), self.m, Q.data_ptr(), self.m, Q.shape[0],
                                                   G.data_ptr(), self.m, self._stream(Q)), "sri_rod_quaternion_rhs")
        return G

    def quaternion_step(self, qe, G, q_base, tips=None, tips_prev=None, update=None):
        """One fused Picard iteration (sri_rod_quaternion_step): returns G_next = 1/2 A(K) (Dnn^-1 G - w q_base^T).

        ``tips`` receives the tip-node values of Q; with ``tips_prev`` and ``update`` (a zeroed int64 CUDA scalar)
        the launch also accumulates the bits of ``max |tips - tips_prev|`` into ``update``."""
        import torch
        Gn = torch.empty_like(G)
        ptr = lambda t: t.data_ptr() if t is not None else None
        _chk(load_library().sri_rod_quaternion_step(self._h, qe.data_ptr(), G.data_ptr(), self.m, q_base.data_ptr(), G.shape[0],
                                                    Gn.data_ptr(), self.m, ptr(tips), ptr(tips_prev), ptr(update),
                                                    self._stream(G)), "sri_rod_quaternion_step")
        return Gn

    def position_rhs(self, Q):
        import torch
        B = torch.empty((Q.shape[0], 3, self.m), dtype=torch.float64, device=Q.device)
        _chk(load_library().sri_rod_position_rhs(self._h, Q.data_ptr(), self.m, Q.shape[0], B.data_ptr(), self.m,
                                                 self._stream(Q)), "sri_rod_position_rhs")
        return B

    def integrate_quaternions(self, qe, q_init=None, tol: float = 1e-14, max_iter: int = 200):
        import torch
        qe = self._check_qe(qe)
        rods = qe.shape[0]
        Q = torch.empty((rods, 4, self.m), dtype=torch.float64, device=qe.device)
        qi = None if q_init is None else (C.c_double * 4)(*[float(v) for v in q_init])
        it, upd = _u32(), C.c_double()
        rc = load_library().sri_rod_integrate_quaternions(self._h, qe.data_ptr(), rods, qi, Q.data_ptr(), self.m, tol, max_iter,
                                                          C.byref(it), C.byref(upd), self._stream(qe))
        self.iterations, self.last_update = int(it.value), float(upd.value)
        _chk(rc, "sri_rod_integrate_quaternions")
        return Q

    def integrate_positions(self, Q, r_init=None):
        import torch
        rods = Q.shape[0]
        R = torch.empty((rods, 3, self.m), dtype=torch.float64, device=Q.device)
        ri = None if r_init is None else (C.c_double * 3)(*[float(v) for v in r_init])
        _chk(load_library().sri_rod_integrate_positions(self._h, Q.data_ptr(), self.m, rods, ri, R.data_ptr(), self.m,
                                                        self._stream(Q)), "sri_rod_integrate_positions")
        return R

    def integrate(self, qe, q_init=None, r_init=None, tol: float = 1e-14, max_iter: int = 200):
        """Quaternions and positions of every rod (sri_rod_integrate, main.cpp:197-201 batched): ``(Q, r)``."""
        import torch
        qe = self._check_qe(qe)
        rods = qe.shape[0]
        Q = torch.empty((rods, 4, self.m), dtype=torch.float64, device=qe.device)
        R = torch.empty((rods, 3, self.m), dtype=torch.float64, device=qe.device)
        qi = None if q_init is None else (C.c_double * 4)(*[float(v) for v in q_init])
        ri = None if r_init is None else (C.c_double * 3)(*[float(v) for v in r_init])
        it, upd = _u32(), C.c_double()
        rc = load_library().sri_rod_integrate(self._h, qe.data_ptr(), rods, qi, ri, Q.data_ptr(), self.m, R.data_ptr(), self.m,
                                              tol, max_iter, C.byref(it), C.byref(upd), self._stream(qe))
        self.iterations, self.last_update = int(it.value), float(upd.value)
        _chk(rc, "sri_rod_integrate")
        return Q, R

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            load_library().sri_rod_destroy(self._h)
            self._h = _vp()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
