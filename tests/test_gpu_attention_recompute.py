"""The recomputing window attention (suhpe_window_attn_fwd_rs / _bwd_rs: row statistics instead of saved probabilities, every chained product
transposed so that no tile goes through LDS) against a float64 statement of src/tiny_vit.py:262-286 and against the round-3 kernels that keep
the probabilities."""
import pytest
import torch

pytestmark = pytest.mark.gpu

CASES = [(7, 6, 3, 28), (14, 12, 2, 14), (7, 18, 5, 7), (5, 2, 2, 10), (8, 3, 1, 16), (13, 4, 2, 13)]


def _ref64(qkv, table, idx, B, H, ws, heads, scale, dout):
    """windows addressed in place in the (B, H, W, C) token tensor; float64 autograd"""
    n = ws * ws
    qkv = qkv.double().cpu().requires_grad_(True)
    table = table.double().cpu().requires_grad_(True)
    x = qkv.view(B, H // ws, ws, H // ws, ws, heads, 3, 32).permute(0, 1, 3, 5, 6, 2, 4, 7).reshape(-1, heads, 3, n, 32)
    q, k, v = x[:, :, 0], x[:, :, 1], x[:, :, 2]
    bias = table[:, idx.cpu().long().view(-1)].view(heads, n, n)
    p = torch.softmax(q @ k.transpose(-1, -2) * scale + bias, dim=-1)
    o = p @ v                                                            # (windows, heads, n, 32)
    out = o.view(B, H // ws, H // ws, heads, ws, ws, 32).permute(0, 1, 4, 2, 5, 3, 6).reshape(B * H * H, heads * 32)
    out.backward(dout.double().cpu())
    return out.detach(), qkv.grad, table.grad


@pytest.mark.parametrize("ws,heads,B,H", CASES)
def test_recomputing_attention_vs_float64_and_saved_probabilities(ws, heads, B, H):
    from semiuhpe_amd import kernels as K
    from semiuhpe_amd.src import tiny_vit as T
    torch.manual_seed(ws * 100 + heads)
    att = T.Attention(heads * 32, 32, heads, ws).cuda()
    att.attention_biases.data.normal_(0, 0.7)
    g = torch.Generator().manual_seed(B + H)
    qkv = (torch.randn(B * H * H, heads * 96, generator=g) * 1.5).cuda()
    dout = torch.randn(B * H * H, heads * 32, generator=g).cuda()
    idx, noff = att._index(qkv.device), att.attention_biases.shape[1]
    inv = att._inverse_index(qkv.device)
    out, stats = K.window_attn_fwd_rs(qkv, att.attention_biases, idx, B, H, H, ws, heads, att.scale)
    out_inf, none = K.window_attn_fwd_rs(qkv, att.attention_biases, idx, B, H, H, ws, heads, att.scale, save_stats=False)
    assert none is None and torch.equal(out, out_inf)
    db = torch.zeros(heads, noff, device="cuda")
    dqkv = K.window_attn_bwd_rs(qkv, out, stats, dout, att.attention_biases, idx, B, H, H, ws, heads, noff, att.scale, db, inverse=inv)
    db_again = torch.zeros(heads, noff, device="cuda")
    dqkv_again = K.window_attn_bwd_rs(qkv, out, stats, dout, att.attention_biases, idx, B, H, H, ws, heads, noff, att.scale, db_again, inverse=inv)
    assert torch.equal(dqkv, dqkv_again) and torch.equal(db, db_again)                      # reproducible bit for bit
    o64, dq64, db64 = _ref64(qkv, att.attention_biases.data, idx, B, H, ws, heads, att.scale, dout)
    rel = lambda a, b: float((a.double().cpu() - b).abs().max() / b.abs().max())
    assert rel(out, o64) < 2e-6
    assert rel(dqkv, dq64) < 5e-6
    assert rel(db, db64) < 5e-6
    # the kernels that keep the probabilities: the same results to rounding (delta comes from dO . O here, from sum P dP there)
    out0, probs = K.window_attn_fwd(qkv, att.attention_biases, idx, B, H, H, ws, heads, att.scale)
    db0 = torch.zeros(heads, noff, device="cuda")
    dq0 = K.window_attn_bwd(qkv, probs, dout, idx, B, H, H, ws, heads, noff, att.scale, db0, inverse=inv)
    assert rel(out, out0.double().cpu()) < 2e-6 and rel(dqkv, dq0.double().cpu()) < 5e-6 and rel(db, db0.double().cpu()) < 5e-6
    # the atomic bias-gradient form accumulates on top of what is there
    dba = db.clone()
    K.window_attn_bwd_rs(qkv, out, stats, dout, att.attention_biases, idx, B, H, H, ws, heads, noff, att.scale, dba, inverse=None)
    assert float((dba - 2 * db).abs().max()) < 3e-5 * float(db.abs().max())


def test_row_statistics_are_the_softmax_normalisers():
    """rowstats[w, h, i] = (max_j s_ij, 1 / sum_j exp(s_ij - max)) of the scaled, biased scores: the contract the backward relies on"""
    from semiuhpe_amd import kernels as K
    from semiuhpe_amd.src import tiny_vit as T
    ws, heads, B, H = 7, 6, 2, 14
    torch.manual_seed(3)
    att = T.Attention(heads * 32, 32, heads, ws).cuda()
    att.attention_biases.data.normal_(0, 0.5)
    qkv = torch.randn(B * H * H, heads * 96, device="cuda")
    idx = att._index(qkv.device)
    out, stats = K.window_attn_fwd_rs(qkv, att.attention_biases, idx, B, H, H, ws, heads, att.scale)
    n = ws * ws
    st = stats.view(-1, heads, 64, 2)[:, :, :n].double().cpu()
    x = qkv.double().cpu().view(B, H // ws, ws, H // ws, ws, heads, 3, 32).permute(0, 1, 3, 5, 6, 2, 4, 7).reshape(-1, heads, 3, n, 32)
    s = x[:, :, 0] @ x[:, :, 1].transpose(-1, -2) * att.scale + att.attention_biases.data.double().cpu()[:, idx.cpu().long().view(-1)].view(heads, n, n)
    m = s.max(-1).values
    assert float((st[..., 0] - m).abs().max()) < 1e-5
    assert float((st[..., 1] * torch.exp(s - m[..., None]).sum(-1) - 1).abs().max()) < 1e-5
    assert float(stats.view(-1, heads, 64, 2)[:, :, n:, 1].abs().max()) == 0.0            # rows beyond the window: weight 0


def test_attention_layer_runs_the_recomputing_kernels(monkeypatch):
    """tiny_vit.Attention keeps (out, rowstats) on the tape -- no N x N tensor -- and gives the gradients of the probability-keeping path"""
    from semiuhpe_amd.src import tiny_vit as T
    assert T.ATTN_RECOMPUTE
    torch.manual_seed(11)
    att = T.Attention(192, 32, 6, 7).cuda().train()
    att.attention_biases.data.normal_(0, 0.3)
    x = torch.randn(4, 14, 14, 192, device="cuda")
    dy = torch.randn(4, 14, 14, 192, device="cuda")

    def run():
        for p in att.parameters():
            p.grad = None
        tape = []
        y = att.fwd(x, tape)
        kept = [t for item in tape if isinstance(item, tuple) for t in item if isinstance(t, tuple)]
        dx = att.bwd(dy, tape)
        return y, dx, {k: p.grad.clone() for k, p in att.named_parameters()}, kept

    y1, dx1, g1, kept = run()
    assert kept and kept[0][1].numel() == 4 * 4 * 6 * 64 * 2                                # row statistics, not 49 x 49 per window and head
    monkeypatch.setattr(T, "ATTN_RECOMPUTE", False)
    y0, dx0, g0, _ = run()
    rel = lambda a, b: float((a - b).abs().max() / b.abs().max())
    assert rel(y1, y0) < 2e-6 and rel(dx1, dx0) < 2e-5
    for k in g0:
        assert rel(g1[k], g0[k]) < 2e-5, k
