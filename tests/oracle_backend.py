"""Adapter that lets the CPU oracle stand behind lsnet_amd.ops for CPU tensors.

TEST INFRASTRUCTURE (also used by bench.py's cpu_baseline leg): implements the same backend
interface as lsnet_amd/ops/hip_backend.py with oracle/oracle_py.py."""
import torch

from oracle import oracle_py as orc


class OracleBackend:
    name = 'oracle'

    @staticmethod
    def _split(om, cfg, weight):
        k2 = 2 * cfg['dg'] * weight.shape[2] * weight.shape[3]
        return om[:, :k2].contiguous(), torch.sigmoid(om[:, k2:]).contiguous()

    def dcn_forward(self, inputs, offsets, masks, weight, bias, cfg, out_hw):
        if cfg.get('fused_om'):
            pairs = [self._split(om, cfg, weight) for om in offsets]
            offsets, masks = [p[0] for p in pairs], [p[1] for p in pairs]
        outs = []
        for i, x in enumerate(inputs):
            sh, sw = cfg['scales'][i]
            out = orc.deform_conv_forward(x, weight, bias, offsets[i], masks[i], cfg['stride'], cfg['pad'],
                                          cfg['dil'], cfg['groups'], cfg['dg'], sh, sw, out_hw=out_hw[i])
            outs.append(out)
        return outs

    def dcn_backward(self, inputs, offsets, masks, weight, grad_outs, cfg, need):
        fused = cfg.get('fused_om')
        if fused:
            pairs = [self._split(om, cfg, weight) for om in offsets]
            offsets, masks = [p[0] for p in pairs], [p[1] for p in pairs]
        gxs, goffs, gmsks = [], [], []
        gw = torch.zeros_like(weight.detach().float().contiguous())
        gb = torch.zeros(weight.shape[0])
        for i, x in enumerate(inputs):
            sh, sw = cfg['scales'][i]
            g = orc.deform_conv_backward(x, weight, offsets[i], masks[i], grad_outs[i], cfg['stride'],
                                         cfg['pad'], cfg['dil'], cfg['groups'], cfg['dg'], sh, sw)
            gxs.append(g['gx']); goffs.append(g['goff']); gmsks.append(g['gmask'])
            gw += g['gw']; gb += g['gb']
        if fused:   # one gradient per level: [d offsets | d mask logits]
            goffs = [torch.cat([go, gm * m * (1 - m)], 1) for go, gm, m in zip(goffs, gmsks, masks)]
            gmsks = [None] * len(goffs)
        return gxs, goffs, gmsks, gw, gb

    def focal_forward(self, logits, targets, gamma, alpha):
        return orc.sigmoid_focal_loss_forward(logits, targets, gamma, alpha)

    def focal_backward(self, logits, targets, d_losses, gamma, alpha):
        return orc.sigmoid_focal_loss_backward(logits, targets, d_losses, gamma, alpha)

    def focal_sum(self, logits, targets, weight, gamma, alpha):
        l = orc.sigmoid_focal_loss_forward(logits, targets, gamma, alpha)
        if weight is not None:
            l = l * weight[:, None]
        return l.sum()

    def focal_backward_weighted(self, logits, targets, weight, scale, gamma, alpha):
        d = torch.ones_like(logits) * scale.reshape(())
        if weight is not None:
            d = d * weight[:, None]
        return orc.sigmoid_focal_loss_backward(logits, targets, d.contiguous(), gamma, alpha)

    def nms(self, dets, iou_thr):
        return orc.nms(dets, iou_thr)
