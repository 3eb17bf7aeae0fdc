"""A differentiable pure-torch deformable convolution (4-tap gather + autograd), written
independently of the oracle; used to pin the oracle's forward and all five gradients."""
import torch


def torch_dcn(x, off, mask, w, b, stride, pad, dil, groups, dg, sh=1.0, sw=1.0):
    B, C, H, W = x.shape
    Co, Cg, kh, kw = w.shape
    Ho, Wo = off.shape[2:]
    K = kh * kw
    ys = torch.arange(Ho).view(1, 1, Ho, 1) * stride - pad
    xs = torch.arange(Wo).view(1, 1, 1, Wo) * stride - pad
    cpg = C // dg
    taps = []
    for k in range(K):
        i, j = divmod(k, kw)
        per_dg = []
        for d in range(dg):
            py = (ys + i * dil).float() * sh + off[:, d * 2 * K + 2 * k:d * 2 * K + 2 * k + 1]
            px = (xs + j * dil).float() * sw + off[:, d * 2 * K + 2 * k + 1:d * 2 * K + 2 * k + 2]
            inside = ((py > -1) & (px > -1) & (py < H) & (px < W)).float()
            y0, x0 = torch.floor(py), torch.floor(px)
            ly, lx = py - y0, px - x0
            xd = x[:, d * cpg:(d + 1) * cpg].reshape(B, cpg, -1)
            val = 0
            for dy, dx, wt in ((0, 0, (1 - ly) * (1 - lx)), (0, 1, (1 - ly) * lx), (1, 0, ly * (1 - lx)),
                               (1, 1, ly * lx)):
                yy, xx = (y0 + dy).long(), (x0 + dx).long()
                ok = ((yy >= 0) & (yy < H) & (xx >= 0) & (xx < W)).float()
                idx = (yy.clamp(0, H - 1) * W + xx.clamp(0, W - 1)).expand(B, cpg, Ho, Wo).reshape(B, cpg, -1)
                val = val + torch.gather(xd, 2, idx).reshape(B, cpg, Ho, Wo) * wt * ok
            val = val * inside
            if mask is not None:
                val = val * mask[:, d * K + k:d * K + k + 1]
            per_dg.append(val)
        taps.append(torch.cat(per_dg, 1))
    col = torch.stack(taps, 2).reshape(B, groups, C // groups * K, Ho * Wo)
    out = torch.einsum('gok,bgkp->bgop', w.reshape(groups, Co // groups, Cg * K), col).reshape(B, Co, Ho, Wo)
    if b is not None:
        out = out + b.view(1, -1, 1, 1)
    return out
