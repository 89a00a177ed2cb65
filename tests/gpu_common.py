"""Shared helpers of the GPU parity tests: call the HIP path through the C ABI and the
oracle (oracle/libpgsd_oracle.so) on the same seeded inputs."""
import ctypes

import numpy as np

import scenario as S

NP_TO_ID = {np.dtype(v): k for k, v in S.NP_TYPES.items()}


def type_id(dt):
    return NP_TO_ID[np.dtype(dt)]


def oracle_pack(src, M, col0=0, out_dtype=None, order=None, bitcast=False):
    """Expected chunk for source array `src` (N_src x stride) via oracle_pack_rows."""
    lib = S.oracle_lib()
    src = np.ascontiguousarray(src)
    if src.ndim == 1:
        src = src.reshape(-1, 1)
    out_dtype = np.dtype(out_dtype or src.dtype)
    N = len(order) if order is not None else src.shape[0]
    dst = np.zeros((N, M), dtype=out_dtype)
    o = np.ascontiguousarray(order, dtype=np.uint32) if order is not None else None
    rc = lib.oracle_pack_rows(dst.ctypes.data, type_id(out_dtype), src.ctypes.data, type_id(src.dtype), N, M,
                              src.shape[1], col0, o.ctypes.data if o is not None else None,
                              1 if bitcast else 0)
    assert rc == 0, rc
    return dst


def hip_pack(jobs, N, stream=None):
    """jobs: list of (dst_tensor, out_np_dtype, M, src_tensor(2-D, full rows), col0, order_tensor, bitcast).

    Calls pgsd_pack_fields (the bare kernel entry of the C ABI) and synchronises."""
    import torch
    from pgsd import _lib
    arr = (_lib.PackJob * len(jobs))()
    for i, (dst, out_dt, M, src, col0, order, bitcast) in enumerate(jobs):
        arr[i].dst = dst.data_ptr()
        arr[i].dst_type = type_id(out_dt)
        arr[i].M = M
        arr[i].src.src = src.data_ptr()
        arr[i].src.order = order.data_ptr() if order is not None else None
        arr[i].src.src_type = type_id(str(src.dtype)[6:])
        arr[i].src.src_stride = src.shape[1] if src.dim() == 2 else 1
        arr[i].src.src_col0 = col0
        arr[i].src.bitcast = 1 if bitcast else 0
    torch.cuda.synchronize()
    rc = _lib.lib.pgsd_pack_fields(len(jobs), arr, N, ctypes.c_void_p(stream) if stream else None, None)
    assert rc == 0, (rc, _lib.last_error())
    torch.cuda.synchronize()


def rand_array(rng, shape, dt):
    dt = np.dtype(dt)
    if dt.kind == 'f':
        a = rng.standard_normal(shape) * 50.0
        return a.astype(dt)
    info = np.iinfo(dt)
    return rng.integers(info.min, info.max, size=shape, dtype=dt, endpoint=True)
