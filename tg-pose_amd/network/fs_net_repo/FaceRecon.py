"""Drop-in for ``network/fs_net_repo/FaceRecon.py``: encoder, topology (PH) predictor, decoder.

The classes keep the reference's attribute names so that ``state_dict()`` keys match a reference
checkpoint exactly (Face_Enc :12, Face_Dec :89, PH_Predictor :120, FaceNet :170).  Called stand-alone,
their forwards run the fused eval-mode HIP pipeline of ``tgpose_amd.engine`` in ``.eval()`` and the
differentiable one of ``tgpose_amd.autograd`` (batch-statistics BatchNorm, dropout, HIP backward) in
``.train()`` -- graph-attached outputs when gradients are being recorded, as in the reference.
"""
import torch
import torch.nn as nn

from ... import engine, ops
from ...config import FLAGS
from . import gcn3d
from .gcn3d import _Packable, _xyz


def _pad_rows(x, C):
    """(B, N, C) rows -> (B, N, FEAT_LD) zero padded (the row stride every layer over the concat buffer reads)"""
    return torch.nn.functional.pad(x.float(), (0, engine.FEAT_LD - C))


class _WithBuffers(_Packable):
    def _sig(self):
        return tuple((p.data_ptr(), p._version) for p in list(self.parameters()) + list(self.buffers()))


class Face_Enc(_WithBuffers):
    def __init__(self):
        super().__init__()
        self.neighbor_num = FLAGS.gcn_n_num
        self.support_num = FLAGS.gcn_sup_num
        self.conv_0 = gcn3d.HSlayer_surface(kernel_num=128, support_num=self.support_num)
        self.conv_1 = gcn3d.HS_layer(128, 128, support_num=self.support_num)
        self.pool_1 = gcn3d.Pool_layer(pooling_rate=4, neighbor_num=4)
        self.conv_2 = gcn3d.HS_layer(128, 256, support_num=self.support_num)
        self.conv_3 = gcn3d.HS_layer(256, 256, support_num=self.support_num)
        self.pool_2 = gcn3d.Pool_layer(pooling_rate=4, neighbor_num=4)
        self.conv_4 = gcn3d.HS_layer(256, 512, support_num=self.support_num)
        self.bn1 = nn.BatchNorm1d(128)
        self.bn2 = nn.BatchNorm1d(256)
        self.bn3 = nn.BatchNorm1d(256)
        # the projection head of feat_global (enable_proj=True; the reference's own callers never pass it)
        self.proj_layer = nn.Sequential(nn.Conv1d(1286, 1286, kernel_size=1, bias=False), nn.BatchNorm1d(1286),
                                        nn.LeakyReLU(negative_slope=0.2), nn.Conv1d(1286, 1286, kernel_size=1, bias=False))

    def project(self, feat_buf):
        """proj_layer on the rows of the encoder's output (FaceRecon.py:80-84, enable_proj=True): feat_buf (B, N, >= 1286 columns:
        the concat buffer in eval mode, the autograd path's feature in training mode) -> feat_global_prj (B, 1286, N)."""
        B, N = feat_buf.shape[:2]
        C = engine.FEAT_C
        if self.training:
            from ... import autograd as tgp_autograd
            pl = self.proj_layer
            x = tgp_autograd.linear(tgp_autograd._pad4(feat_buf[:, :, :C].reshape(B * N, C)), tgp_autograd._pad4(tgp_autograd._w2(pl[0])))
            x = tgp_autograd.bn_act(x, pl[1], act=1, slope=0.2)
            y = tgp_autograd.linear(tgp_autograd._pad4(x), tgp_autograd._pad4(tgp_autograd._w2(pl[3])))
            return y.view(B, N, C).permute(0, 2, 1)
        dev = feat_buf.device
        sd = {"proj_layer." + k: v for k, v in self.proj_layer.state_dict().items()}
        po = engine.proj_operands({"encoder." + k: v for k, v in sd.items()}, "", dev)
        scale, shift = engine._fold_values(engine._dev_sd({"encoder." + k: v for k, v in sd.items()}, dev), po["name"])
        pad = engine.PROJ_LD - C
        h = ops.linear_rows(feat_buf.reshape(B * N, feat_buf.shape[2]), po["w1"][:, : feat_buf.shape[2]].contiguous(),
                            scale=torch.cat([scale, torch.ones(pad, device=dev)]), shift=torch.cat([shift, torch.zeros(pad, device=dev)]),
                            act=1, slope=0.2)
        y = ops.linear_rows(h, po["w2"])
        return y.view(B, N, -1)[:, :, :C].permute(0, 2, 1)

    def forward(self, vertices, cat_id, enable_proj=False):
        """vertices (B,N,3) already centred, cat_id (B,1) -> feat (B,N,1286), feat_global (B,1286,N) [through proj_layer when
        enable_proj, FaceRecon.py:80-84]."""
        dev = vertices.device
        if self.training:
            from ... import autograd as tgp_autograd
            xyz = _xyz(vertices)
            feat, _parts = tgp_autograd.encoder(self, xyz, cat_id.to(dev), engine.draw_sample_idx(xyz.shape[1]),
                                                tgp_autograd._GraphSource(dev, None, None, ""), self.neighbor_num, FLAGS.obj_c)
            feat = feat[:, :, : engine.FEAT_C]
            return feat, (self.project(feat) if enable_proj else feat.permute(0, 2, 1))
        conv = self._packed(lambda: engine.pack_encoder(engine._dev_sd(self.state_dict(), dev), "", dev))
        pk = type("EncPack", (), dict(conv=conv))
        xyz = vertices.detach().float().contiguous()
        feat, _ = engine.encoder_forward(pk, xyz, cat_id, engine.draw_sample_idx(xyz.shape[1]),
                                         engine.Graphs(dev), self.neighbor_num, FLAGS.obj_c)
        prj = self.project(feat) if enable_proj else None
        feat = feat[:, :, : engine.FEAT_C]
        return feat, (prj if enable_proj else feat.permute(0, 2, 1))


class Face_Dec(_WithBuffers):
    def __init__(self, dim_fuse):
        super().__init__()
        self.recon_num = 3
        self.conv1d_block = nn.Sequential(
            nn.Conv1d(dim_fuse, 512, 1), nn.BatchNorm1d(512), nn.ReLU(inplace=True),
            nn.Conv1d(512, 512, 1), nn.BatchNorm1d(512), nn.ReLU(inplace=True),
            nn.Conv1d(512, 256, 1), nn.BatchNorm1d(256), nn.ReLU(inplace=True))
        self.recon_head = nn.Sequential(nn.Conv1d(256, 128, 1), nn.BatchNorm1d(128), nn.ReLU(inplace=True),
                                        nn.Conv1d(128, self.recon_num, 1))

    def forward(self, x):
        """x (B,1286,N) -> recon (B,N,3)"""
        B, C, N = x.shape
        dev = x.device
        if self.training:
            from ... import autograd as tgp_autograd
            return tgp_autograd.decoder(self, _pad_rows(x.transpose(1, 2), C), None)
        dec, dec_out = self._packed(lambda: engine.pack_decoder(engine._dev_sd(self.state_dict(), dev), ""))
        pk = type("DecPack", (), dict(dec=dec, dec_out=dec_out))
        rows = torch.zeros(B, N, engine.FEAT_LD, device=dev, dtype=torch.float32)
        rows[:, :, :C].copy_(x.detach().float().transpose(1, 2))
        return engine.decoder_forward(pk, rows, None, N)


class PH_Predictor(_WithBuffers):
    def __init__(self):
        super().__init__()
        self.output_channels = FLAGS.output_channels
        self.conv_5 = nn.Sequential(nn.Conv1d(128 + 128 + 256 + 256 + 512 + 6, 1024, kernel_size=1, bias=False),
                                    nn.BatchNorm1d(1024), nn.LeakyReLU(negative_slope=0.2))
        self.linear1 = nn.Linear(1024 * 2, 1024, bias=False)
        self.bn5 = nn.BatchNorm1d(1024)
        self.dp1 = nn.Dropout(p=0.5)
        self.linear2 = nn.Linear(1024, self.output_channels)
        self.linear3 = nn.Linear(1024, self.output_channels)
        self.linear4 = nn.Linear(self.output_channels, 1286)
        self.linear5 = nn.Linear(self.output_channels, 1286)
        self.ac2 = nn.Sigmoid()
        self.ac3 = nn.Sigmoid()

    def forward(self, feat):
        """feat (B,N,1286) -> feat + back-projected topology code (B,1286,N), h1, h2 (B,2500)."""
        B, N, C = feat.shape
        dev = feat.device
        if self.training:
            from ... import autograd as tgp_autograd
            back, h1, h2 = tgp_autograd.ph_predictor(self, _pad_rows(feat, C))
            return feat.permute(0, 2, 1) + back[:, :C].unsqueeze(-1), h1, h2
        ph = self._packed(lambda: engine.pack_ph(engine._dev_sd(self.state_dict(), dev), ""))
        pk = type("PhPack", (), dict(ph=ph))
        rows = torch.zeros(B, N, engine.FEAT_LD, device=dev, dtype=torch.float32)
        rows[:, :, :C].copy_(feat.detach().float())
        h1, h2, back = engine.ph_forward(pk, rows, N)
        return feat.permute(0, 2, 1) + back[:, :C].unsqueeze(-1), h1, h2


class FaceNet(_WithBuffers):
    def __init__(self):
        super().__init__()
        self.encoder = Face_Enc()
        self.decoder = Face_Dec(1286)
        self.ph_pred = PH_Predictor()

    def forward(self, vertices, cat_id, enable_proj=False, pred_PH=True):
        """-> recon (B,N,3), feat (B,N,1286), feat_global (B,1286,N) [through the encoder's proj_layer when enable_proj], h1, h2
        (FaceRecon.py:178-200)"""
        dev = vertices.device
        if self.training:
            from ... import autograd as tgp_autograd
            xyz = _xyz(vertices)
            featp, parts = tgp_autograd.encoder(self.encoder, xyz, cat_id.to(dev), engine.draw_sample_idx(xyz.shape[1]),
                                                tgp_autograd._GraphSource(dev, None, None, "encoder."), self.encoder.neighbor_num,
                                                FLAGS.obj_c)
            # the layers over `feat` as PoseNet9D.forward runs them: factored over the up-sampling when the encoder hands out the
            # operands (autograd.FACTORED), else over the padded concat buffer
            x5 = xd = None
            if parts is not None:
                dec0 = self.decoder.conv1d_block[0]
                x5, xd = tgp_autograd.feat_consumers_factored(parts, [(tgp_autograd._w2(self.ph_pred.conv_5[0]), None),
                                                                      (tgp_autograd._w2(dec0), dec0.bias)])
            h1 = h2 = back = None
            if pred_PH:
                back, h1, h2 = tgp_autograd.ph_predictor(self.ph_pred, featp, x5)
            recon = tgp_autograd.decoder(self.decoder, featp, back, xd)
            f = featp[:, :, : engine.FEAT_C]
            return recon, f, (self.encoder.project(f) if enable_proj else f.permute(0, 2, 1)), h1, h2
        pk = self._packed(lambda: engine.Packed(self.state_dict(), dev, face="", with_heads=False))
        xyz = vertices.detach().float().contiguous()
        N = xyz.shape[1]
        feat, _ = engine.encoder_forward(pk, xyz, cat_id, engine.draw_sample_idx(N), engine.Graphs(dev),
                                         self.encoder.neighbor_num, FLAGS.obj_c)
        h1 = h2 = back = None
        if pred_PH:
            h1, h2, back = engine.ph_forward(pk, feat, N)
        recon = engine.decoder_forward(pk, feat, back, N)
        f = feat[:, :, : engine.FEAT_C]
        return recon, f, (self.encoder.project(feat) if enable_proj else f.permute(0, 2, 1)), h1, h2
