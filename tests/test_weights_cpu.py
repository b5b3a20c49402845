"""torchvision state_dict (.pth) loader (reference call site utils_cv/action_recognition/model.py:421)."""
import numpy as np
import pytest
import torch

from flickering_adversarial_video_amd import videoresnet_spec as vs


@pytest.mark.parametrize("arch", ["mc3_18", "r2plus1d_18"])
def test_pth_state_dict_round_trip(tmp_path, arch):
    W = vs.synthetic_weights(arch, 3)
    sd = {k: torch.from_numpy(v) for k, v in W.items()}
    for k in list(sd):                                   # what a real BatchNorm3d state_dict also carries
        if k.endswith("running_var"):
            sd[k.replace("running_var", "num_batches_tracked")] = torch.tensor(7)
    torch.save(sd, tmp_path / "a.pth")
    got = vs.load_weights(str(tmp_path / "a.pth"), arch)
    assert set(got) == set(W) and all(np.array_equal(got[k], W[k]) and got[k].dtype == np.float32 for k in W)
    if arch != "mc3_18":
        return
    # nn.DataParallel prefix + checkpoint wrapper + fp16 storage
    torch.save({"state_dict": {"module." + k: v.half() if v.dtype == torch.float32 else v for k, v in sd.items()}, "epoch": 3}, tmp_path / "b.pt")
    got = vs.load_weights(str(tmp_path / "b.pt"), arch)
    assert set(got) == set(W) and np.allclose(got["fc.weight"], W["fc.weight"], atol=1e-3)
    np.savez(tmp_path / "c.npz", **W)
    got = vs.load_weights(str(tmp_path / "c.npz"), arch)
    assert all(np.array_equal(got[k], W[k]) for k in W)


def test_wrong_architecture_is_refused(tmp_path):
    torch.save({k: torch.from_numpy(v) for k, v in vs.synthetic_weights("mc3_18", 1).items()}, tmp_path / "mc3.pth")
    with pytest.raises((KeyError, ValueError)):
        vs.load_weights(str(tmp_path / "mc3.pth"), "r2plus1d_18")
    with pytest.raises(ValueError):
        vs.load_weights(str(tmp_path / "mc3.pth"), "r3d_18")           # same names as mc3_18, other kernel shapes from layer2 on
