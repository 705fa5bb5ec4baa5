"""Pins the oracle's ResNet body (oracle/detr_oracle.py::resnet_body) against an INDEPENDENT implementation of the same
published architecture: Hugging Face `transformers.ResNetModel` (bottleneck, stride on the 3x3 = torchvision's v1.5,
depths [3,4,6,3] / [3,4,23,3]).  The reference builds its backbone from torchvision (src/models/backbone.py:59-94:
`getattr(torchvision.models, name)(..., norm_layer=FrozenBatchNorm2d)` + IntermediateLayerGetter(layer4)); torchvision is
not in this image, `transformers` is, and its ResNet shares no code with ours.  The same weights go into both through the
torchvision key names the reference's checkpoints carry; the layer4 feature maps must agree to fp32 reorder noise.

This does not make the pin a reference run (only torchvision itself would), but both sides of the backbone comparison are
no longer the builder's code."""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

transformers = pytest.importorskip("transformers")


def hf_resnet(sd, depths, prefix="backbone.0.body."):
    """transformers.ResNetModel carrying the torchvision-named weights of `sd` (BatchNorm in eval mode == FrozenBatchNorm2d,
    eps 1e-5 on both sides: src/models/backbone.py:46-57)"""
    from transformers import ResNetConfig, ResNetModel
    cfg = ResNetConfig(num_channels=3, embedding_size=64, hidden_sizes=[256, 512, 1024, 2048], depths=list(depths),
                       layer_type="bottleneck", hidden_act="relu", downsample_in_first_stage=False,
                       downsample_in_bottleneck=False)
    m = ResNetModel(cfg).eval()
    new = {}

    def bn(dst, src):
        for k in ("weight", "bias", "running_mean", "running_var"):
            new[dst + "normalization." + k] = sd[prefix + src + "." + k]

    new["embedder.embedder.convolution.weight"] = sd[prefix + "conv1.weight"]
    bn("embedder.embedder.", "bn1")
    for s, n in enumerate(depths):
        for i in range(n):
            src = f"layer{s + 1}.{i}."
            dst = f"encoder.stages.{s}.layers.{i}."
            for j in range(3):
                new[dst + f"layer.{j}.convolution.weight"] = sd[prefix + src + f"conv{j + 1}.weight"]
                bn(dst + f"layer.{j}.", src + f"bn{j + 1}")
            if i == 0:
                new[dst + "shortcut.convolution.weight"] = sd[prefix + src + "downsample.0.weight"]
                bn(dst + "shortcut.", src + "downsample.1")
    missing, unexpected = m.load_state_dict(new, strict=False)
    assert not unexpected, unexpected
    assert all(k.endswith("num_batches_tracked") for k in missing), missing
    return m


@pytest.mark.parametrize("backbone,depths,scene", [("resnet50", (3, 4, 6, 3), True), ("resnet50", (3, 4, 6, 3), False),
                                                   ("resnet101", (3, 4, 23, 3), True)])
def test_oracle_resnet_body_matches_transformers_resnet(backbone, depths, scene):
    import detr_oracle
    from odam_amd import weights
    torch.manual_seed(0)
    sd = weights.make_state_dict(seed=0, backbone=backbone, scene=scene)
    x = torch.randn(2, 3, 160, 224)
    with torch.no_grad():
        ours = detr_oracle.resnet_body(x, sd, blocks=depths)
        theirs = hf_resnet(sd, depths)(x).last_hidden_state
    assert ours.shape == theirs.shape == (2, 2048, 5, 7)
    scale = theirs.abs().max().item()
    assert scale > 0
    err = (ours - theirs).abs().max().item()
    assert err <= 2e-5 * scale, (err, scale)
    # every stage, not only the end: hidden states of the independent model against a staged run of the oracle
    with torch.no_grad():
        hs = hf_resnet(sd, depths)(x, output_hidden_states=True).hidden_states      # embedder output, then the four stages
    assert len(hs) == 5 and hs[1].shape[1] == 256 and hs[4].shape[1] == 2048
    assert torch.equal(hs[4], theirs)
