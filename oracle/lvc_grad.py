"""numpy restatement of the location-variable convolution AND its gradients -- TEST INFRASTRUCTURE.

Reference: TimeAware_LVCBlock.location_variable_convolution, modules/FastDiff/module/modules.py:220-253, dilation = 1 (its only call
site passes 1, modules.py:216): pad -> unfold(hop + 2 pad, hop) -> unfold(kernel_size, 1) -> einsum('bildsk,biokl->bolsd') + bias.
Restated as   out[b,o,l*hop+s] = bias[b,o,l] + sum_{i,k} xpad[b,i,l*hop+s+k] * K[b,i,o,k,l]   (xpad = x zero-padded by (ks-1)/2),
with the gradients written out by hand.  Pinned on the reference function and on torch.autograd through it
(tests/golden/lvc_grad.npz, oracle/gen_golden.py gen_lvc_grad).
"""
import numpy as np


def _frames(x, ks, T, hop):
    """[B,Cin,L] -> windows [B,Cin,T,hop,ks] of the zero-padded signal."""
    pad = (ks - 1) // 2
    xp = np.pad(x, ((0, 0), (0, 0), (pad, pad)))
    idx = (np.arange(T)[:, None, None] * hop + np.arange(hop)[None, :, None] + np.arange(ks)[None, None, :])
    return xp[:, :, idx]


def lvc_forward(x, K, bias, hop):
    B, Cin, L = x.shape
    _, _, Cout, ks, T = K.shape
    assert L == T * hop
    win = _frames(x, ks, T, hop)                                   # b i l s k
    out = np.einsum("bilsk,biokl->bols", win, K) + bias[:, :, :, None]
    return out.reshape(B, Cout, L)


def lvc_backward(x, K, dout, hop):
    """-> dx [B,Cin,L], dK [B,Cin,Cout,ks,T], dbias [B,Cout,T]."""
    B, Cin, L = x.shape
    _, _, Cout, ks, T = K.shape
    pad = (ks - 1) // 2
    win = _frames(x, ks, T, hop)
    d = dout.reshape(B, Cout, T, hop)
    dK = np.einsum("bols,bilsk->biokl", d, win)
    dbias = d.sum(-1)
    dwin = np.einsum("bols,biokl->bilsk", d, K)                    # gradient of every window element
    dxp = np.zeros((B, Cin, L + 2 * pad), dout.dtype)
    idx = (np.arange(T)[:, None, None] * hop + np.arange(hop)[None, :, None] + np.arange(ks)[None, None, :])
    np.add.at(dxp, (slice(None), slice(None), idx), dwin)
    return dxp[:, :, pad:pad + L], dK, dbias
