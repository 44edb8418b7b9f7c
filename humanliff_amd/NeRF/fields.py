"""Host-side mirror of human_diffusion/NeRF/fields.py (only what the render path touches).

PositionalEncoding exists for state_dict compatibility (`view_enc._freqs`, `view_enc._phases`
buffers, fields.py:58-66); the encoding itself is evaluated inside the HIP ray-march kernel.
Unlike the reference, importing this module does NOT call
torch.autograd.set_detect_anomaly(True) (fields.py:2).
"""
import math

import numpy as np
import torch


class PositionalEncoding(torch.nn.Module):
    def __init__(self, num_freqs=6, d_in=3, freq_factor=np.pi, include_input=True):
        super().__init__()
        self.num_freqs = num_freqs
        self.d_in = d_in
        self.include_input = include_input
        self.d_out = num_freqs * 2 * d_in + (d_in if include_input else 0)
        octave = torch.pow(2.0, torch.arange(num_freqs, dtype=torch.float32))
        self.register_buffer("_freqs", octave.repeat_interleave(2).view(1, -1, 1))
        ph = torch.zeros(2 * num_freqs)
        ph[1::2] = math.pi * 0.5
        self.register_buffer("_phases", ph.view(1, -1, 1))


def img2mse(x, y):
    return torch.mean((x - y) ** 2)


def mse2psnr(x):
    return -10.0 * torch.log(x) / torch.log(torch.tensor([10.0], device=x.device))


def to8b(x):
    return (255 * np.clip(x, 0, 1)).astype(np.uint8)
