"""The library's own radix sort (csrc/radix_sort.hip: the sort behind voxel keys, the grid build's cell ids) against
torch.sort(stable=True): keys AND the order of equal keys' values, ragged sizes, every pass count, heavy duplicates."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def sp():
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a HIP device (no CPU fallback exists)")
    import sycl_points_amd.api as api

    return api


def run_sort(L, sp, keys, vals, bits):
    n = keys.numel()
    ka, va = keys.clone(), vals.clone()
    kb, vb = torch.empty_like(ka), torch.empty_like(va)
    ws = torch.empty(max(int(L.sp_internal_radix_sort_workspace_bytes(n)), 16), dtype=torch.uint8, device="cuda")
    in_b = C.c_int(-1)
    sp.check(L.sp_internal_radix_sort_u32(sp._ptr(ka), sp._ptr(kb), sp._ptr(va), sp._ptr(vb), n, bits, sp._ptr(ws), ws.numel(),
                                          C.byref(in_b), sp._stream()))
    torch.cuda.synchronize()
    return (kb, vb) if in_b.value == 1 else (ka, va)


@pytest.mark.parametrize("n,bits", [(1, 8), (63, 5), (4096, 8), (4097, 9), (100_003, 17), (1_000_000, 23), (1_000_000, 24),
                                    (262_144, 32), (3_000_001, 22)])
def test_radix_sort_matches_stable_torch_sort(sp, n, bits):
    L = sp._lib.lib()
    g = torch.Generator(device="cuda").manual_seed(n + bits)
    hi = (1 << bits) - 1
    keys = torch.randint(0, min(hi, 2**31 - 1) + 1, (n,), generator=g, device="cuda", dtype=torch.int64)
    if bits == 32:
        keys = keys * 2 + torch.randint(0, 2, (n,), generator=g, device="cuda", dtype=torch.int64)
    vals = torch.arange(n, device="cuda", dtype=torch.int32)
    ku = (keys & 0xFFFFFFFF).to(torch.int64)
    k_i32 = torch.where(ku >= 2**31, ku - 2**32, ku).to(torch.int32)  # same bits as the unsigned key
    sk, sv = run_sort(L, sp, k_i32, vals, bits)
    ref_k, ref_order = torch.sort(ku, stable=True)
    got_k = sk.to(torch.int64) & 0xFFFFFFFF
    assert torch.equal(got_k, ref_k)
    assert torch.equal(sv.to(torch.int64), ref_order)  # stability: equal keys keep their input order


def test_radix_sort_heavy_duplicates_and_presorted(sp):
    L = sp._lib.lib()
    n = 500_000
    keys = (torch.arange(n, device="cuda") % 7).to(torch.int32)  # seven distinct keys: whole tiles of one digit
    vals = torch.arange(n, device="cuda", dtype=torch.int32)
    sk, sv = run_sort(L, sp, keys, vals, 3)
    ref_k, ref_order = torch.sort(keys.to(torch.int64), stable=True)
    assert torch.equal(sk.to(torch.int64), ref_k) and torch.equal(sv.to(torch.int64), ref_order)
    keys = torch.arange(n, device="cuda", dtype=torch.int32)  # already sorted, 19 bits
    sk, sv = run_sort(L, sp, keys, vals, 19)
    assert torch.equal(sk, keys) and torch.equal(sv, vals)
