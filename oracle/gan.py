"""Oracle: HiFi-GAN and Fre-GAN generator forward (fp32, ATen CPU kernels).

`w` is a dict of FOLDED weights: '<conv>.weight', '<conv>.bias' with the
reference's module names (what remove_weight_norm() leaves,
models/vocoder/hifigan/models.py:152-162).  `h` is the json config dict.
"""
import torch
import torch.nn.functional as F

LRELU_SLOPE = 0.1  # hifigan/models.py:8, fregan/generator.py:8


def get_padding(kernel_size, dilation=1):  # utils/util.py:60-61
    return int((kernel_size * dilation - dilation) / 2)


def fold_weight_norm_state(state):
    """weight_g/weight_v -> weight (torch._weight_norm, dim=0), other keys kept."""
    out = {}
    for k, v in state.items():
        if k.endswith(".weight_g"):
            p = k[: -len(".weight_g")]
            g, vv = v.float(), state[p + ".weight_v"].float()
            out[p + ".weight"] = torch._weight_norm(vv, g, 0)
        elif k.endswith(".weight_v"):
            continue
        else:
            out[k] = v.float()
    return out


def _resblock1(w, prefix, x, kernel_size, dilations):
    # ResBlock1.forward hifigan/models.py:37-44 == fregan/generator.py:41-48
    for d, dil in enumerate(dilations):
        xt = F.leaky_relu(x, LRELU_SLOPE)
        xt = F.conv1d(xt, w[f"{prefix}.convs1.{d}.weight"], w[f"{prefix}.convs1.{d}.bias"],
                      dilation=dil, padding=get_padding(kernel_size, dil))
        xt = F.leaky_relu(xt, LRELU_SLOPE)
        xt = F.conv1d(xt, w[f"{prefix}.convs2.{d}.weight"], w[f"{prefix}.convs2.{d}.bias"],
                      dilation=1, padding=get_padding(kernel_size, 1))
        x = xt + x
    return x


def _resblock2(w, prefix, x, kernel_size, dilations):
    # ResBlock2.forward hifigan/models.py:63-68 == fregan/generator.py:67-72 == sublayer/vits_modules.py:236-245 (x_mask=None):
    # two convs, the first two dilations of the block
    for d in range(2):
        xt = F.leaky_relu(x, LRELU_SLOPE)
        xt = F.conv1d(xt, w[f"{prefix}.convs.{d}.weight"], w[f"{prefix}.convs.{d}.bias"],
                      dilation=dilations[d], padding=get_padding(kernel_size, dilations[d]))
        x = xt + x
    return x


def _resblock(w, h, prefix, x, kernel_size, dilations):
    # resblock = ResBlock1 if h.resblock == '1' else ResBlock2  (hifigan/models.py:100, fregan/generator.py:91, vits.py:251)
    if str(h.get("resblock", "1")) == "1":
        return _resblock1(w, prefix, x, kernel_size, dilations)
    return _resblock2(w, prefix, x, kernel_size, dilations)


def _up(w, name, x, u):
    # ConvTranspose1d(k, u, padding=u//2+u%2, output_padding=u%2) models.py:120-123
    return F.conv_transpose1d(x, w[name + ".weight"], w[name + ".bias"], stride=u,
                              padding=u // 2 + u % 2, output_padding=u % 2)


def _up_interp(w, name, x, u, k):
    # h.sampling_rate == 24000: Sequential(InterpolationBlock(u), Conv1d(k, padding=(k-1)//2))
    # models.py:74-91,107-118.  interpolate(size=T*u, mode='nearest') repeats every sample u times.
    x = F.interpolate(x, size=x.shape[-1] * u, mode="nearest")
    return F.conv1d(x, w[name + ".1.weight"], w[name + ".1.bias"], padding=(k - 1) // 2)


def hifigan_forward(w, h, mel):
    """Generator.forward, models/vocoder/hifigan/models.py:134-150.  mel [B,80,F] -> [B,1,F*hop]
    (24 kHz variant: one sample less per even-kernel upsampling stage)."""
    nk = len(h["resblock_kernel_sizes"])
    interp = int(h.get("sampling_rate", 16000)) == 24000
    x = F.conv1d(mel, w["conv_pre.weight"], w["conv_pre.bias"], padding=3)
    for i, u in enumerate(h["upsample_rates"]):
        x = F.leaky_relu(x, LRELU_SLOPE)
        x = _up_interp(w, f"ups.{i}", x, u, h["upsample_kernel_sizes"][i]) if interp else _up(w, f"ups.{i}", x, u)
        xs = None
        for j in range(nk):
            r = _resblock(w, h, f"resblocks.{i * nk + j}", x, h["resblock_kernel_sizes"][j],
                          h["resblock_dilation_sizes"][j])
            xs = r if xs is None else xs + r
        x = xs / nk
    x = F.leaky_relu(x)  # default slope 0.01 (models.py:146)
    x = F.conv1d(x, w["conv_post.weight"], w["conv_post.bias"], padding=3)
    return torch.tanh(x)


def fregan_forward(w, h, mel, top_k=4):
    """FreGAN.forward, models/vocoder/fregan/generator.py:137-166."""
    nk = len(h["resblock_kernel_sizes"])
    rates = h["upsample_rates"]
    n_up = len(rates)
    cond_level = n_up - top_k  # generator.py:89
    x = F.conv1d(mel, w["conv_pre.weight"], w["conv_pre.bias"], padding=3)
    output = None
    for i in range(n_up):
        if i >= cond_level:
            mel = _up(w, f"cond_up.{i - cond_level}", mel, rates[i - 1])  # generator.py:111-118
            x = x + mel
        if i > cond_level:
            src = x if output is None else output
            src = F.interpolate(src, scale_factor=rates[i], mode="nearest")  # nn.Upsample generator.py:106
            n = f"res_output.{i - cond_level - 1}.1"
            output = F.conv1d(src, w[n + ".weight"], w[n + ".bias"])
        x = F.leaky_relu(x, LRELU_SLOPE)
        x = _up(w, f"ups.{i}", x, rates[i])
        xs = None
        for j in range(nk):
            r = _resblock(w, h, f"resblocks.{i * nk + j}", x, h["resblock_kernel_sizes"][j],
                          h["resblock_dilation_sizes"][j])
            xs = r if xs is None else xs + r
        x = xs / nk
        if output is not None:
            output = output + x
    x = F.leaky_relu(output)
    x = F.conv1d(x, w["conv_post.weight"], w["conv_post.bias"], padding=3)
    return torch.tanh(x)


def vits_generator_forward(w, h, x, g=None):
    """VITS decoder: Generator.forward, models/synthesizer/models/vits.py:273-291 (ResBlock1,
    sublayer/vits_modules.py:180-216 with x_mask=None).  x [B, initial_channel, T] latent, g [B, gin, 1]
    speaker embedding or None -> [B, 1, T*hop].  Differences from the HiFi-GAN generator: conv_pre is not
    weight-normed, optional `x + cond(g)` after it, ConvTranspose1d padding (k-u)//2 (:255-257), conv_post
    has no bias."""
    nk = len(h["resblock_kernel_sizes"])
    x = F.conv1d(x, w["conv_pre.weight"], w["conv_pre.bias"], padding=3)
    if g is not None:
        x = x + F.conv1d(g, w["cond.weight"], w["cond.bias"])
    for i, (u, k) in enumerate(zip(h["upsample_rates"], h["upsample_kernel_sizes"])):
        x = F.leaky_relu(x, LRELU_SLOPE)
        x = F.conv_transpose1d(x, w[f"ups.{i}.weight"], w[f"ups.{i}.bias"], stride=u, padding=(k - u) // 2)
        xs = None
        for j in range(nk):
            r = _resblock(w, h, f"resblocks.{i * nk + j}", x, h["resblock_kernel_sizes"][j], h["resblock_dilation_sizes"][j])
            xs = r if xs is None else xs + r
        x = xs / nk
    x = F.leaky_relu(x)
    x = F.conv1d(x, w["conv_post.weight"], None, padding=3)
    return torch.tanh(x)
