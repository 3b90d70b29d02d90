"""Host-side data path: vectorised masking against the oracle's restatement of the reference loop."""
import torch

from oracle import visualbert_oracle as vo
from visualbert_amd.data import mask_tokens


def test_mask_tokens_matches_reference_loop_draw_for_draw():
    g = torch.Generator().manual_seed(7)
    B, T, V, MASK = 6, 40, 30522, 103
    ids = torch.randint(1000, V, (B, T), generator=g, dtype=torch.int64)
    maskable = torch.ones(B, T, dtype=torch.bool)
    maskable[:, 0] = False                      # [CLS]
    maskable[:, -3:] = False                    # [SEP] + padding
    u = torch.rand(B, T, generator=g, dtype=torch.float64)
    u[0, 5] = 0.15 * 0.8                        # boundaries of the three outcomes, exercised exactly
    u[0, 6] = 0.15 * 0.9
    u[0, 7] = 0.15
    r = torch.randint(0, V, (B, T), generator=g, dtype=torch.int64)
    out, labels = mask_tokens(ids, maskable, V, MASK, uniforms=u, random_ids=r)
    for b in range(B):
        idx = torch.nonzero(maskable[b]).reshape(-1).tolist()       # the reference only sees the real tokens
        toks, labs = vo.random_word_loop([int(ids[b, i]) for i in idx], [float(u[b, i]) for i in idx],
                                         [int(r[b, i]) for i in idx], MASK)
        assert [int(out[b, i]) for i in idx] == toks
        assert [int(labels[b, i]) for i in idx] == labs
        rest = [i for i in range(T) if i not in idx]
        assert all(int(out[b, i]) == int(ids[b, i]) and int(labels[b, i]) == -1 for i in rest)


def test_mask_tokens_rates():
    g = torch.Generator().manual_seed(11)
    ids = torch.randint(1000, 30522, (64, 128), generator=g, dtype=torch.int64)
    out, labels = mask_tokens(ids, torch.ones_like(ids, dtype=torch.bool), 30522, 103, generator=g)
    sel = labels != -1
    rate = sel.float().mean().item()
    assert abs(rate - 0.15) < 0.015
    masked = (out == 103) & sel
    kept = (out == ids) & sel
    assert abs(masked.sum().item() / sel.sum().item() - 0.8) < 0.03
    assert abs(kept.sum().item() / sel.sum().item() - 0.1) < 0.03
    assert torch.equal(labels[sel], ids[sel]) and torch.equal(out[~sel], ids[~sel])
