"""clt_mse_bf16_colsum_noalloc_f32 — the entry point for callers that cannot allocate (library-owned fixed scratch,
resolved per device with hipGetSymbolAddress) — against clt_mse_bf16_colsum_f32 through the C ABI: bit-identical
loss sum, bf16 gradient rows and column sums, on two calls in a row, the second on a smaller input (stale partials of
the first call stay in the scratch), and a clear error beyond the fixed maximum."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _both(rows, n, seed, stride_pad=0):
    from crosslayer_transcoder_amd._lib import lib
    g = torch.Generator().manual_seed(seed)
    r = torch.randn(rows, n, generator=g).cuda()
    ybuf = torch.randn(rows, 2, n + stride_pad, generator=g).cuda()          # y = batch[:, 1]: strided rows
    y = ybuf[:, 1, :n]
    yrs = ybuf.stride(0)
    mean, std = torch.randn(n, generator=g).cuda(), (torch.rand(n, generator=g) + 0.5).cuda()
    gs = 2.0 / (rows * n)
    st = torch.cuda.current_stream().cuda_stream
    outs = []
    for noalloc in (False, True):
        g16 = torch.zeros(rows, n, dtype=torch.bfloat16, device="cuda")
        cs = torch.full((n,), 7.0, device="cuda")
        ls = torch.zeros(1, dtype=torch.float64, device="cuda")
        if noalloc:
            rc = lib.clt_mse_bf16_colsum_noalloc_f32(r.data_ptr(), y.data_ptr(), yrs, mean.data_ptr(), std.data_ptr(), rows, n, gs,
                                                     g16.data_ptr(), cs.data_ptr(), 0, ls.data_ptr(), st)
        else:
            nb = lib.clt_mse_bf16_colsum_workspace_bytes(rows, n)
            ws = torch.empty(nb, dtype=torch.uint8, device="cuda")
            rc = lib.clt_mse_bf16_colsum_f32(r.data_ptr(), y.data_ptr(), yrs, mean.data_ptr(), std.data_ptr(), rows, n, gs,
                                             g16.data_ptr(), cs.data_ptr(), 0, ls.data_ptr(), ws.data_ptr(), nb, st)
        assert rc == 0, rc
        torch.cuda.synchronize()
        outs.append((g16.cpu(), cs.cpu(), ls.cpu()))
    # and against plain arithmetic (fp64), so that "equal" is not "equally wrong"
    d = r.double().cpu() - (y.double().cpu() - mean.double().cpu()) / std.double().cpu()
    assert abs(float(outs[1][2]) - float((d.float() ** 2).double().sum())) <= 1e-6 * float((d ** 2).sum())
    assert torch.allclose(outs[1][1].double(), (gs * d).sum(0), rtol=1e-4, atol=1e-6 * float((gs * d).abs().max()) * rows)
    return outs


def test_noalloc_entry_point_matches_workspace_entry_point():
    from crosslayer_transcoder_amd._lib import lib
    for rows, n, seed, pad in ((1000, 9216, 1, 0), (96, 512, 2, 8)):        # two calls in a row, the second smaller
        (g0, c0, l0), (g1, c1, l1) = _both(rows, n, seed, pad)
        assert torch.equal(g0.view(torch.int16), g1.view(torch.int16)), "bf16 gradient rows differ"
        assert torch.equal(c0, c1), "column sums differ"
        assert torch.equal(l0, l1), "loss sum differs"
    # beyond the fixed scratch: a clear error, nothing launched
    cap = lib.clt_mse_bf16_colsum_noalloc_max_bytes()
    rows, n = 16384, 9216
    assert lib.clt_mse_bf16_colsum_workspace_bytes(rows, n) > cap
    t = torch.zeros(8, device="cuda")
    rc = lib.clt_mse_bf16_colsum_noalloc_f32(t.data_ptr(), t.data_ptr(), n, t.data_ptr(), t.data_ptr(), rows, n, 1.0, t.data_ptr(),
                                             t.data_ptr(), 0, t.data_ptr(), torch.cuda.current_stream().cuda_stream)
    assert rc == -3          # CLT_ERR_WORKSPACE
