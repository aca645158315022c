"""The RCCL gather of the C ABI (include/pds_amd.h, "Multi-GPU") through ctypes, with the one rank a
one-GPU box allows: communicator from a unique id and from pds_comm_init_all, equal and ragged row
counts (world 1: a device copy through ncclAllGather), in-place use, the statistics all-reduce."""
import ctypes

import numpy as np
import pytest

from pydrobert_speech_amd import _native

pytestmark = pytest.mark.gpu


def _counts(values):
    return (ctypes.c_int64 * len(values))(*values)


def test_gather_rows_single_rank():
    import torch

    lib = _native.lib()
    torch.cuda.set_device(0)
    ident = ctypes.create_string_buffer(128)
    _native.check(lib.pds_comm_unique_id(ident), "pds_comm_unique_id")
    assert any(ident.raw)
    comm = ctypes.c_void_p()
    _native.check(lib.pds_comm_init_rank(ident, 1, 0, ctypes.byref(comm)), "pds_comm_init_rank")
    try:
        assert lib.pds_comm_world(comm) == 1 and lib.pds_comm_rank(comm) == 0
        stream = torch.cuda.current_stream().cuda_stream
        local = torch.randn(1000, 40, device="cuda")
        out = torch.zeros(1000, 40, device="cuda")
        _native.check(lib.pds_gather_rows(comm, local.data_ptr(), _counts([1000]), 160, out.data_ptr(), stream),
                      "pds_gather_rows")
        torch.cuda.synchronize()
        assert torch.equal(out, local)
        # float64 rows (CMVN output), in place: the rank's slice of d_out is its d_local
        both = torch.randn(77, 81, device="cuda", dtype=torch.float64)
        keep = both.clone()
        _native.check(lib.pds_gather_rows(comm, both.data_ptr(), _counts([77]), 81 * 8, both.data_ptr(), stream),
                      "pds_gather_rows")
        table = torch.arange(2 * 41, dtype=torch.float64, device="cuda")
        _native.check(lib.pds_allreduce_sum_f64(comm, table.data_ptr(), table.numel(), stream), "pds_allreduce")
        torch.cuda.synchronize()
        assert torch.equal(both, keep) and torch.equal(table.cpu(), torch.arange(2 * 41, dtype=torch.float64))
        # nothing to move; bad arguments
        assert lib.pds_gather_rows(comm, None, _counts([0]), 160, None, stream) == 0
        assert lib.pds_gather_rows(comm, local.data_ptr(), _counts([-1]), 160, out.data_ptr(), stream) == -1
        assert b"negative" in lib.pds_last_error()
        assert lib.pds_gather_rows(comm, local.data_ptr(), _counts([5]), 0, out.data_ptr(), stream) == -1
        assert lib.pds_gather_rows(None, local.data_ptr(), _counts([5]), 4, out.data_ptr(), stream) == -1
    finally:
        lib.pds_comm_destroy(comm)


def test_comm_init_all_one_device():
    import torch

    lib = _native.lib()
    comms = (ctypes.c_void_p * 1)()
    _native.check(lib.pds_comm_init_all(1, None, comms), "pds_comm_init_all")
    try:
        assert lib.pds_comm_world(comms[0]) == 1 and lib.pds_comm_rank(comms[0]) == 0
        stream = torch.cuda.current_stream().cuda_stream
        local = torch.arange(12, dtype=torch.float32, device="cuda").reshape(3, 4)
        out = torch.empty_like(local)
        assert lib.pds_comm_group_start() == 0
        rc = lib.pds_gather_rows(comms[0], local.data_ptr(), _counts([3]), 16, out.data_ptr(), stream)
        assert lib.pds_comm_group_end() == 0
        _native.check(rc, "pds_gather_rows")
        torch.cuda.synchronize()
        assert np.array_equal(out.cpu().numpy(), np.arange(12, dtype=np.float32).reshape(3, 4))
    finally:
        lib.pds_comm_destroy(comms[0])
    assert lib.pds_comm_init_all(0, None, comms) == -1
