"""DeepSDF decoder module (architecture of facebookresearch/DeepSDF as used by the reference,
core/graph/deep_sdf_decoder.py:18-111). Holds parameters with the reference's state-dict keys
(lin{l}.weight / .bias, or .weight_g / .weight_v under weight_norm) so checkpoints load unchanged;
`inference` is a plain PyTorch evaluation used for validation only -- the renderer never calls it,
it packs the parameters for the MFMA kernels (distr.decoder_pack)."""
import torch
import torch.nn as nn


class Decoder(nn.Module):
    def __init__(self, latent_size, dims, last_dim=1, dropout=None, dropout_prob=0.0, norm_layers=(), latent_in=(),
                 weight_norm=False, xyz_in_all=None, use_tanh=False, latent_dropout=False):
        super(Decoder, self).__init__()
        widths = [latent_size + 3] + list(dims) + [last_dim]
        self.num_layers = len(widths)
        self.norm_layers, self.latent_in = norm_layers, latent_in
        self.latent_dropout, self.xyz_in_all = latent_dropout, xyz_in_all
        self.weight_norm, self.use_tanh = weight_norm, use_tanh
        self.dropout, self.dropout_prob = dropout, dropout_prob
        for l in range(self.num_layers - 1):
            n_out = widths[l + 1]
            if (l + 1) in latent_in:
                n_out -= widths[0]                      # room for the re-injected [latent|xyz]
            elif xyz_in_all and l != self.num_layers - 2:
                n_out -= 3
            lin = nn.Linear(widths[l], n_out)
            if weight_norm and l in norm_layers:
                lin = nn.utils.weight_norm(lin)
            setattr(self, 'lin%d' % l, lin)
            if (not weight_norm) and norm_layers is not None and l in norm_layers:
                setattr(self, 'bn%d' % l, nn.LayerNorm(n_out))
        self.th = nn.Tanh()

    def inference(self, inp):
        xyz = inp[:, -3:]
        if inp.shape[1] > 3 and self.latent_dropout:        # deep_sdf_decoder.py:84-89 (the identity in eval mode)
            inp = torch.cat([nn.functional.dropout(inp[:, :-3], p=0.2, training=self.training), xyz], 1)
        x = inp
        last = self.num_layers - 2
        for l in range(self.num_layers - 1):
            if l in self.latent_in:
                x = torch.cat([x, inp], 1)
            elif l != 0 and self.xyz_in_all:
                x = torch.cat([x, xyz], 1)
            x = getattr(self, 'lin%d' % l)(x)
            if l == last and self.use_tanh:
                x = torch.tanh(x)
            if l < last:
                if hasattr(self, 'bn%d' % l):
                    x = getattr(self, 'bn%d' % l)(x)
                x = torch.relu(x)
                if self.dropout is not None and l in self.dropout:
                    x = nn.functional.dropout(x, p=self.dropout_prob, training=self.training)
        return self.th(x)

    forward = inference
