"""create_mlp with the reference's initialisers (modules/util.py:4-79)."""
import numpy as np
import torch


def _create_mlp(input_w, output_w, num_layers, hidden_w=128, bias=True):
    if num_layers == 0:
        return torch.nn.Sequential(torch.nn.Identity())
    if num_layers == 1:
        return torch.nn.Sequential(torch.nn.Linear(input_w, output_w, bias=bias))
    layers = [torch.nn.Linear(input_w, hidden_w)]
    for _ in range(num_layers - 2):
        layers += [torch.nn.ReLU(inplace=True), torch.nn.Linear(hidden_w, hidden_w)]
    layers += [torch.nn.ReLU(inplace=True), torch.nn.Linear(hidden_w, output_w, bias=bias)]
    return torch.nn.Sequential(*layers)


def create_mlp(input_w, output_w, num_layers, hidden_w=128, skip=None, initializer=None, bias=True, **kwargs):
    if skip is not None:
        raise NotImplementedError("skip connections are not used by this config")
    net = _create_mlp(input_w, output_w, num_layers, hidden_w, bias)

    def init(m):
        if isinstance(m, torch.nn.Linear):
            if initializer == "kaiming":
                torch.nn.init.kaiming_uniform_(m.weight)
            elif initializer == "xavier":
                torch.nn.init.xavier_uniform_(m.weight, gain=np.sqrt(2))
            elif initializer == "xavier_sigmoid":
                torch.nn.init.xavier_uniform_(m.weight, gain=1)
            else:
                return
            if m.bias is not None:
                torch.nn.init.constant_(m.bias, 0)

    net.apply(init)
    return net
