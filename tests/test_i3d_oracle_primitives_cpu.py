"""Known-answer checks of the I3D oracle's building blocks against DIRECT evaluations of the documented TensorFlow-1.15 /
dm-sonnet-1.23 semantics (the packages themselves cannot be imported here), written as plain numpy loops that share no
code with oracle/i3d_ref.py:

* ``SAME`` padding of a strided window op: out = ceil(n / s); total pad = max((out - 1) * s + k - n, 0); pad_before = total // 2,
  the extra cell goes AFTER (tf.nn.convolution / max_pool3d documentation);
* snt.Conv3D(use_bias=False) + snt.BatchNorm(scale=False, eps=1e-3) in inference mode + relu (i3d.py:51-71);
* tf.nn.max_pool3d(SAME): padded cells never win;
* the Logits endpoint: 2x7x7 VALID average pool, 1x1x1 convolution with bias, mean over the remaining time axis (i3d.py:459-474)."""
import numpy as np
import torch

from oracle import i3d_ref

P = i3d_ref.PREFIX


def same_pad_doc(n, k, s):
    out = -(-n // s)
    total = max((out - 1) * s + k - n, 0)
    return out, total // 2


def conv_same_loops(x, w, s):
    """x [T,H,W,Cin], w [kt,kh,kw,Cin,Cout] -> [To,Ho,Wo,Cout]"""
    k = w.shape[:3]
    (To, pt), (Ho, ph), (Wo, pw) = (same_pad_doc(n, kk, ss) for n, kk, ss in zip(x.shape[:3], k, s))
    y = np.zeros((To, Ho, Wo, w.shape[4]))
    for ot in range(To):
        for oh in range(Ho):
            for ow in range(Wo):
                for dt in range(k[0]):
                    for dh in range(k[1]):
                        for dw in range(k[2]):
                            t, h, ww = ot * s[0] - pt + dt, oh * s[1] - ph + dh, ow * s[2] - pw + dw
                            if 0 <= t < x.shape[0] and 0 <= h < x.shape[1] and 0 <= ww < x.shape[2]:
                                y[ot, oh, ow] += x[t, h, ww] @ w[dt, dh, dw]
    return y


def pool_same_loops(x, k, s):
    (To, pt), (Ho, ph), (Wo, pw) = (same_pad_doc(n, kk, ss) for n, kk, ss in zip(x.shape[:3], k, s))
    y = np.full((To, Ho, Wo, x.shape[3]), -np.inf)
    for ot in range(To):
        for oh in range(Ho):
            for ow in range(Wo):
                for dt in range(k[0]):
                    for dh in range(k[1]):
                        for dw in range(k[2]):
                            t, h, ww = ot * s[0] - pt + dt, oh * s[1] - ph + dh, ow * s[2] - pw + dw
                            if 0 <= t < x.shape[0] and 0 <= h < x.shape[1] and 0 <= ww < x.shape[2]:
                                y[ot, oh, ow] = np.maximum(y[ot, oh, ow], x[t, h, ww])
    return y


def test_same_padding_rule():
    for n, k, s in ((224, 7, 2), (64, 7, 2), (112, 3, 2), (56, 3, 2), (28, 3, 2), (16, 2, 2), (14, 3, 1), (7, 3, 2), (9, 2, 2), (5, 7, 2)):
        out, before = same_pad_doc(n, k, s)
        b, a = i3d_ref.same_pad(n, k, s)
        assert b == before and (out - 1) * s + k <= n + b + a and b <= a <= b + 1, (n, k, s, b, a)
    assert i3d_ref.same_pad(224, 7, 2) == (2, 3)          # the stem: the extra cell is AFTER


def test_unit3d_against_direct_loops():
    rng = np.random.default_rng(0)
    for shape, k, s in (((6, 5, 8), (7, 7, 7), (2, 2, 2)), ((3, 4, 5), (3, 3, 3), (1, 1, 1)), ((2, 3, 3), (1, 1, 1), (1, 1, 1))):
        cin, cout = 2, 3
        x = rng.standard_normal((*shape, cin))
        w = rng.standard_normal((*k, cin, cout))
        mean, var, beta = rng.standard_normal(cout), rng.uniform(0.5, 1.5, cout), rng.standard_normal(cout)
        ref = np.maximum((conv_same_loops(x, w, s) - mean) / np.sqrt(var + 1e-3) + beta, 0.0)
        W = {P + "u/conv_3d/w": torch.from_numpy(w), P + "u/batch_norm/moving_mean": torch.from_numpy(mean.reshape(1, 1, 1, 1, -1)),
             P + "u/batch_norm/moving_variance": torch.from_numpy(var.reshape(1, 1, 1, 1, -1)), P + "u/batch_norm/beta": torch.from_numpy(beta.reshape(1, 1, 1, 1, -1))}
        got = i3d_ref.unit3d(torch.from_numpy(x).permute(3, 0, 1, 2)[None], W, "u", k, s)[0].permute(1, 2, 3, 0).numpy()
        np.testing.assert_allclose(got, ref, rtol=1e-10, atol=1e-10)


def test_maxpool_same_against_direct_loops():
    rng = np.random.default_rng(1)
    for shape, k, s in (((4, 7, 6), (1, 3, 3), (1, 2, 2)), ((5, 7, 6), (3, 3, 3), (2, 2, 2)), ((4, 6, 6), (2, 2, 2), (2, 2, 2)),
                        ((3, 5, 4), (3, 3, 3), (1, 1, 1))):
        x = rng.standard_normal((*shape, 3)) - 2.0          # mostly negative: a zero-padded pool would let the padding win
        got = i3d_ref.maxpool_same(torch.from_numpy(x).permute(3, 0, 1, 2)[None], k, s)[0].permute(1, 2, 3, 0).numpy()
        np.testing.assert_array_equal(got, pool_same_loops(x, k, s))


def test_logits_head_against_direct_evaluation():
    """a network input is not needed: feed the head's input (the Mixed_5c map [1,1024,8,7,7]) through the oracle's own tail by
    evaluating the documented formula directly and comparing with i3d_logits on a full (tiny-T) forward's last endpoint"""
    rng = np.random.default_rng(2)
    feat = rng.standard_normal((8, 7, 7, 1024))
    w, b = rng.standard_normal((1024, 400)) * 0.01, rng.standard_normal(400)
    pooled = np.stack([feat[t:t + 2].mean(axis=(0, 1, 2)) for t in range(7)])          # 2x7x7 VALID, stride 1 -> [7,1024]
    ref = (pooled @ w + b).mean(axis=0)
    x = torch.from_numpy(feat).permute(3, 0, 1, 2)[None]
    y = torch.nn.functional.avg_pool3d(x, (2, 7, 7), (1, 1, 1))
    W = {P + "Logits/Conv3d_0c_1x1/conv_3d/w": torch.from_numpy(w.reshape(1, 1, 1, 1024, 400)), P + "Logits/Conv3d_0c_1x1/conv_3d/b": torch.from_numpy(b)}
    got = i3d_ref.unit3d(y, W, "Logits/Conv3d_0c_1x1", (1, 1, 1), bn=False, relu=False, bias=True).mean(dim=2).reshape(-1).numpy()
    np.testing.assert_allclose(got, ref, rtol=1e-10, atol=1e-10)
    # ... and that is exactly what i3d_logits does after Mixed_5c (read from its source so that a change there is noticed)
    import inspect
    src = inspect.getsource(i3d_ref.i3d_logits)
    assert "avg_pool3d" in src and "(2, 7, 7)" in src and "Logits/Conv3d_0c_1x1" in src and "mean" in src
