"""Drop-in counterpart of the reference's vae/conditional_prior_network.py: the network that predicts
p(z | context) (ConditionalPriorNetwork, reference :25-209) and the Gaussian-Gaussian KL (:212-233).

Like the sub-modules of CVAEMemRand, the torch.nn layers are PARAMETER CONTAINERS with the reference's
names and shapes; forward() runs on the gfx950 kernels through the owning model.  The stack is the
context encoder's (conv embedding, LSTM, compression Linear) followed by two Linear heads with no
non-linearity in between, so compression and heads fold into one Linear(mem_hidden -> 2L) and the whole
network runs in the posterior encoder's slot of a second packed weight blob (vsv_encode, which = 0).
"""
from __future__ import annotations

from collections import OrderedDict

import torch
import torch.nn as nn


class ConditionalPriorNetwork(nn.Module):
    def __init__(self, config: dict):
        super().__init__()
        self.config = config
        self.device = config["device"]
        L = config["latent_dim"]
        fh, fw = config["feat_dim"]
        hidden = config["ctx_surface_hidden"]
        layers = OrderedDict()
        if config["use_dense_surface"]:                      # Flatten + Linear+ReLU stack (reference :90-99)
            cin = fh * fw
            layers["flatten"] = nn.Flatten()
            for i, cout in enumerate(hidden):
                layers[f"prior_enc_dense_{i}"] = nn.Linear(cin, cout)
                layers[f"prior_enc_activation_{i}"] = nn.ReLU()
                cin = cout
            width = cin
        else:
            cin = 1
            for i, cout in enumerate(hidden):
                layers[f"prior_enc_conv_{i}"] = nn.Conv2d(cin, cout, kernel_size=3, stride=1, padding=config["padding"])
                layers[f"prior_enc_activation_{i}"] = nn.ReLU()
                cin = cout
            layers["flatten"] = nn.Flatten()
            width = cin * fh * fw
        self.surface_embedding = nn.Sequential(layers)
        if config["ex_feats_dim"] > 0:
            if config["ctx_ex_feats_hidden"] is not None:
                raise NotImplementedError("ctx_ex_feats_hidden must be None")
            self.ex_feats_embedding = nn.Identity()
            width += config["ex_feats_dim"]
        n_int = config["interaction_layers"]
        if n_int is not None and n_int > 0:
            inter = OrderedDict()
            for i in range(n_int):
                inter[f"prior_interact_linear_{i}"] = nn.Linear(width, width)
                inter[f"prior_interaction_activation_{i}"] = nn.ReLU()
            inter["prior_interact_linear_final"] = nn.Linear(width, width)
            self.interaction = nn.Sequential(inter)
        else:
            self.interaction = nn.Identity()
        kind = {"lstm": nn.LSTM, "gru": nn.GRU}.get(config["mem_type"], nn.RNN)
        self.mem = kind(input_size=width, hidden_size=config["mem_hidden"], num_layers=config["mem_layers"],
                        batch_first=True, dropout=config["mem_dropout"])
        if config["compress_context"]:
            self.compression = nn.Linear(config["mem_hidden"], L)
            hid = L
        else:
            self.compression = nn.Identity()
            hid = config["mem_hidden"]
        self.prior_mean_layer = nn.Linear(hid, L)
        self.prior_logvar_layer = nn.Linear(hid, L)

    def forward(self, context: dict):
        """context {"surface": (B,C,H,W)[, "ex_feats": (B,C,D)]} -> (prior_mean, prior_logvar), each (B,C,L)."""
        ref = self.__dict__.get("_vsv_owner")
        owner = ref() if ref is not None else None
        if owner is None:
            raise RuntimeError("this prior network is not attached to a CVAEMemRandConditionalPrior")
        return owner._prior_forward(context)


def kl_divergence_gaussians(mu_q, logvar_q, mu_p, logvar_p):
    """KL(q || p) between diagonal Gaussians: sum over latent dims, mean over batch and time (reference :212-233)."""
    kl = 0.5 * (logvar_p - logvar_q + torch.exp(logvar_q - logvar_p)
                + ((mu_q - mu_p) ** 2) / torch.exp(logvar_p) - 1)
    return kl.sum(dim=-1).mean()
