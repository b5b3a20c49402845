"""Generate golden vectors from the REFERENCE's own importable attack classes.

Run in the build container only (needs /root/reference):   python tests/golden/make_golden.py
Writes tests/golden/torch_attack_golden.npz.  The fixture holds data only (seeded inputs and the
reference's outputs); no reference source travels.

Import recipe: the reference's utils_cv/action_recognition/model.py imports
torchvision / decord / IPython at module scope; inert stubs are registered for those three missing
third-party modules, nothing in the reference is modified.  ``Losses.__init__`` hard-codes
``.to('cuda')`` (model.py:146), so the instance is built with ``object.__new__`` and its attributes
set by hand exactly as ``__init__`` would (model.py:142-166).
"""
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "torch_attack_golden.npz")


def import_reference_model():
    sys.dont_write_bytecode = True
    sys.path.insert(0, REF)

    def stub(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    class Compose:
        def __init__(self, t):
            self.transforms = t

        def __call__(self, x):
            for t in self.transforms:
                x = t(x)
            return x

    tv = stub("torchvision")
    tv.transforms = stub("torchvision.transforms", Compose=Compose)
    tv.models = stub("torchvision.models")
    tv.models.video = stub("torchvision.models.video")
    tv.models.video.resnet = stub("torchvision.models.video.resnet", VideoResNet=object)
    stub("decord", VideoReader=object)
    ip = stub("IPython")
    ip.display = stub("IPython.display")
    import utils_cv.action_recognition.model as m
    return m


def make_losses(m, beta_1, lambda_, margin, improve_loss, logits, attack_type, targeted=False, target_class=None):
    L = object.__new__(m.Losses)
    L.beta_1, L.lambda_, L.targeted, L.target_class = beta_1, lambda_, targeted, target_class
    L.margin = torch.FloatTensor([margin])
    L.logits, L.attack_type = logits, attack_type
    L.adv_loss = L.improve_adversarial_loss if improve_loss else L.ce_adversarial_loss
    L.regularization_loss = (L.flickering_regularization_loss if attack_type == "flickering"
                             else L.L12_regularization_loss)
    return L


def main():
    m = import_reference_model()
    rng = np.random.default_rng(20240517)
    G = {}

    # ---- 1. Perturbation.forward: flicker [3,16,1,1] and dense [3,16,8,8], max_norm 0.1 / 0.2 ----
    x = (rng.standard_normal((2, 3, 16, 8, 8)) * 1.4 + 0.3).astype(np.float32)   # spans the clamp range
    w = rng.standard_normal(x.shape).astype(np.float32)
    G["pert_x"], G["pert_w"] = x, w
    for tag, size, mn in (("flk01", (3, 16, 1, 1), 0.1), ("flk02", (3, 16, 1, 1), 0.2), ("dense02", (3, 16, 8, 8), 0.2)):
        d = rng.uniform(-0.3, 0.3, size).astype(np.float32)
        P = m.Perturbation(size, device="cpu", max_norm=mn)
        P.init_perturbation(d)
        xt = torch.from_numpy(x)
        out = P.forward([xt, True])
        (out * torch.from_numpy(w)).sum().backward()
        G[f"pert_{tag}_delta"] = d
        G[f"pert_{tag}_max_norm"] = np.float32(mn)
        G[f"pert_{tag}_xadv"] = out.detach().numpy()
        G[f"pert_{tag}_grad"] = P.perturbation.grad.numpy().copy()
        G[f"pert_{tag}_clean"] = P.forward([xt, False]).numpy()
        th, ro = P.metric_calc()
        G[f"pert_{tag}_metric"] = np.array([th.item(), ro.item()], np.float32)
        c, raw = P.get_perturbation()
        G[f"pert_{tag}_clamped"] = c.detach().numpy()
        G[f"pert_{tag}_zero_one"] = np.asarray(P.apply_perturbation(xt))            # model.py:103-112: numpy [B,T,H,W,3] in [0,1]
    G["pert_min_value"] = np.float64(m.Perturbation((3, 2, 1, 1), device="cpu").min_value)
    G["pert_max_value"] = np.float64(m.Perturbation((3, 2, 1, 1), device="cpu").max_value)

    # ---- 2. Losses.__call__: {improve-prob, improve-logits, CE} x {flickering, L12} ----
    logits = (rng.standard_normal((4, 400)) * 3).astype(np.float32)
    labels = logits.argmax(1).copy()
    labels[1] = (labels[1] + 7) % 400      # a non-argmax (already-fooled) case
    labels[3] = int(np.argsort(logits[3])[-2])  # runner-up: small positive margin
    G["loss_logits"], G["loss_labels"] = logits, labels.astype(np.int64)
    for dtag, size in (("flk", (3, 16, 1, 1)), ("dense", (3, 16, 8, 8))):
        d = rng.uniform(-0.15, 0.15, size).astype(np.float32)
        G[f"loss_{dtag}_delta"] = d
        for mode, improve, use_logits in (("improve_prob", True, False), ("improve_logits", True, True), ("ce", False, False)):
            atype = "flickering" if dtag == "flk" else "L12"
            L = make_losses(m, 0.5, 1.0, 0.05, improve, use_logits, atype)
            lg = torch.from_numpy(logits).requires_grad_(True)
            dt = torch.from_numpy(d).requires_grad_(True)
            prob = torch.softmax(lg, 1)
            loss, adv, reg = L(torch.from_numpy(labels.astype(np.int64)), lg, prob, dt)
            loss.backward()
            key = f"loss_{dtag}_{mode}"
            G[key + "_out"] = np.array([loss.item(), adv.item(), reg.item()], np.float64)
            G[key + "_dlogits"] = lg.grad.numpy().copy()
            G[key + "_ddelta"] = dt.grad.numpy().copy()
            G[key + "_label_prob"] = L.label_prob.detach().numpy().reshape(-1)
    # a second beta_1 / lambda_ / margin setting
    L = make_losses(m, 0.3, 2.5, 0.1, True, False, "flickering")
    lg = torch.from_numpy(logits).requires_grad_(True)
    dt = torch.from_numpy(G["loss_flk_delta"]).requires_grad_(True)
    loss, adv, reg = L(torch.from_numpy(labels.astype(np.int64)), lg, torch.softmax(lg, 1), dt)
    loss.backward()
    G["loss_alt_out"] = np.array([loss.item(), adv.item(), reg.item()], np.float64)
    G["loss_alt_dlogits"] = lg.grad.numpy().copy()
    G["loss_alt_ddelta"] = dt.grad.numpy().copy()

    # ---- 3. Adversarial_metrics ----
    M = m.Adversarial_metrics()
    clean = (rng.standard_normal((6, 400)) * 2).astype(np.float32)
    gt = clean.argmax(1).copy()
    gt[0] = (gt[0] + 3) % 400                          # clean-misclassified video
    adv_lg = clean.copy()
    adv_lg[2, (gt[2] + 1) % 400] += 50                 # fooled
    adv_lg[4, (gt[4] + 9) % 400] += 50                 # fooled
    adv_lg[0, gt[0]] += 50                             # adversarial input "fixes" a misclassified clip
    miss, valid = M.accuracy_for_eval(torch.from_numpy(adv_lg), torch.from_numpy(gt.astype(np.int64)),
                                      clean_pred=torch.from_numpy(clean))
    G["met_clean"], G["met_adv"], G["met_gt"] = clean, adv_lg, gt.astype(np.int64)
    G["met_miss_valid"] = np.array([miss.item(), valid.item()], np.float64)
    d = rng.uniform(-0.2, 0.2, (3, 16, 1, 1)).astype(np.float32)
    th, ro = M.adversarial_metric(torch.from_numpy(d))
    G["met_delta"], G["met_thick_rough"] = d, np.array([th.item(), ro.item()], np.float32)
    # Adversarial_metrics.accuracy (model.py:262-291): top-1 fooling percentage, untargeted (needs clean_pred) and targeted.  (topk with
    # maxk > 1 raises in the reference under torch >= 1.5 -- .view(-1) of a transposed product, model.py:287 -- so only the default
    # topk=(1,) has reference outputs to pin.)
    acc = M.accuracy(torch.from_numpy(adv_lg), torch.from_numpy(gt.astype(np.int64)), clean_pred=torch.from_numpy(clean))
    G["met_accuracy_top1"] = np.array([float(a) for a in acc], np.float64)
    tc = int((gt[2] + 1) % 400)
    Mt = m.Adversarial_metrics(targeted=True, target_class=tc)
    acc_t = Mt.accuracy(torch.from_numpy(adv_lg), torch.from_numpy(gt.astype(np.int64)))
    G["met_accuracy_targeted_class"] = np.int64(tc)
    G["met_accuracy_targeted"] = np.array([float(a) for a in acc_t], np.float64)

    # ---- 4. 20-step mini attack on a tiny seeded conv3d net, loop ordering of model.py:1056-1101 ----
    torch.manual_seed(7)
    net = torch.nn.Sequential(torch.nn.Conv3d(3, 8, 3, padding=1), torch.nn.ReLU(),
                              torch.nn.Conv3d(8, 8, 3, padding=1), torch.nn.ReLU(),
                              torch.nn.AdaptiveAvgPool3d(1), torch.nn.Flatten(), torch.nn.Linear(8, 400)).eval()
    for p in net.parameters():
        p.requires_grad_(False)
    G["mini_w1"], G["mini_b1"] = net[0].weight.numpy().copy(), net[0].bias.numpy().copy()
    G["mini_w2"], G["mini_b2"] = net[2].weight.numpy().copy(), net[2].bias.numpy().copy()
    G["mini_fw"], G["mini_fb"] = net[6].weight.numpy().copy(), net[6].bias.numpy().copy()
    xm = (rng.standard_normal((1, 3, 16, 6, 6)) * 1.2).astype(np.float32)
    d0 = rng.uniform(-0.005, 0.005, (3, 16, 1, 1)).astype(np.float32)    # model.py:946-948
    P = m.Perturbation((3, 16, 1, 1), device="cpu", max_norm=0.2)
    P.init_perturbation(d0.copy())      # from_numpy shares memory; Adam updates in place
    L = make_losses(m, 0.5, 1.0, 0.05, True, True, "flickering")        # r2plus1d single-video settings
    opt = torch.optim.Adam(P.parameters(), lr=1e-3)                       # model.py:868
    xt = torch.from_numpy(xm)
    tgt = net(xt).argmax(1)
    traj, losses = [], []
    for _ in range(20):
        out = net(P.forward([xt, True]))
        sc = torch.softmax(out, 1)
        pc, _ = P.get_perturbation()
        loss, adv, reg = L(tgt, out, sc, pc)
        loss.backward()
        opt.step()
        opt.zero_grad()
        traj.append(P.perturbation.detach().numpy().copy())
        losses.append([loss.item(), adv.item(), reg.item()])
    G["mini_x"], G["mini_delta0"], G["mini_target"] = xm, d0, tgt.numpy()
    G["mini_traj"], G["mini_losses"] = np.stack(traj), np.array(losses, np.float64)

    np.savez_compressed(OUT, **G)
    print("wrote", OUT, os.path.getsize(OUT), "bytes;", len(G), "arrays")


if __name__ == "__main__":
    main()
