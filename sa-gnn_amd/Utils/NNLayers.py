"""Parameter registry and small layer helpers with the reference's names
(reference Utils/NNLayers.py). Parameters are fp32 device tensors; on the hot path the
activation and residual adds are fused into the SpMM kernel, so `Activate` / `FC` here serve the
callers around the path (prediction head, SSL meta-net) and keep the L2 registry identical —
including the dead [d, d] weight every messagePropagate call registers (reference model.py:81).
"""
from __future__ import annotations

import math

import torch

paramId = 0
mhsaId = 0                       # MultiHeadSelfAttention instances since reset(): names their variables
params: dict[str, torch.Tensor] = {}
regParams: dict[str, torch.Tensor] = {}
leaky = 0.1                      # reference NNLayers.py:10; model.prepareModel sets it from args
_device = torch.device("cpu")
_generator: torch.Generator | None = None


def reset(device="cpu", seed: int = 100):
    """Clears the registry (the reference relies on a fresh process per run) and seeds the
    initialiser stream (reference main.py:21-23 seeds everything with 100)."""
    global paramId, mhsaId, _device, _generator
    paramId = 0
    mhsaId = 0
    params.clear()
    regParams.clear()
    _device = torch.device(device)
    _generator = torch.Generator(device="cpu")
    _generator.manual_seed(seed)


def getParamId():
    global paramId
    paramId += 1
    return paramId


def getMhsaId():
    """Creation-order index of a MultiHeadSelfAttention (TF names them dense, dense_1, ... per graph);
    restarts with reset() so a rebuilt model gets the names its checkpoint holds."""
    global mhsaId
    mhsaId += 1
    return mhsaId


def getParam(name):
    return params[name]


def xavier_uniform_(shape):
    """tf.contrib.layers.xavier_initializer(uniform=True): U(-l, l), l = sqrt(6/(fan_in+fan_out));
    for rank > 2 the leading dims multiply both fans."""
    shape = tuple(int(s) for s in (shape if hasattr(shape, "__len__") else (shape,)))
    if len(shape) == 1:
        fan_in = fan_out = shape[0]
    else:
        rf = 1
        for s in shape[:-2]:
            rf *= s
        fan_in, fan_out = shape[-2] * rf, shape[-1] * rf
    lim = math.sqrt(6.0 / (fan_in + fan_out))
    t = torch.rand(shape, generator=_generator, dtype=torch.float32) * (2 * lim) - lim
    return t.to(_device)


def defineParam(name, shape, dtype=torch.float32, reg=False, initializer="xavier", trainable=True):
    assert name not in params, "name %s already exists" % name      # reference NNLayers.py:46
    shape = tuple(int(s) for s in (shape if hasattr(shape, "__len__") else (shape,)))
    if initializer == "xavier":
        ret = xavier_uniform_(shape)
    elif initializer == "zeros":
        ret = torch.zeros(shape, dtype=dtype, device=_device)
    elif initializer == "ones":
        ret = torch.ones(shape, dtype=dtype, device=_device)
    elif isinstance(initializer, torch.Tensor):
        ret = initializer.to(device=_device, dtype=dtype).clone()
    else:
        raise ValueError("ERROR: Unrecognized initializer")
    ret.requires_grad_(bool(trainable))
    params[name] = ret
    if reg:
        regParams[name] = ret
    return ret


def defineRandomNameParam(shape, dtype=torch.float32, reg=False, initializer="xavier", trainable=True):
    return defineParam("defaultParamName%d" % getParamId(), shape, dtype, reg, initializer, trainable)


def getOrDefineParam(name, shape, dtype=torch.float32, reg=False, initializer="xavier",
                     trainable=True, reuse=False):
    if name in params:
        assert reuse, "Reusing Param %s Not Specified" % name           # reference NNLayers.py:74
        if reg and name not in regParams:
            regParams[name] = params[name]
        return params[name]
    return defineParam(name, shape, dtype, reg, initializer, trainable)


def Bias(data, name=None, reg=False, reuse=False, initializer="zeros"):
    temName = name if name is not None else "defaultParamName%d" % getParamId()
    bias = getOrDefineParam(temName + "Bias", data.shape[-1], reg=False, initializer=initializer, reuse=reuse)
    if reg:
        regParams[temName + "Bias"] = bias
    return data + bias


def ActivateHelp(data, method):
    if method == "relu":
        return torch.relu(data)
    if method == "sigmoid":
        return torch.sigmoid(data)
    if method == "tanh":
        return torch.tanh(data)
    if method == "leakyRelu":
        return torch.maximum(leaky * data, data)                        # reference NNLayers.py:136
    raise Exception("Error Activation Function")


def Activate(data, method, useBN=False):
    if useBN:
        raise NotImplementedError("BN is unused by the reference's model path")
    return ActivateHelp(data, method)


def FC(inp, outDim, name=None, useBias=False, activation=None, reg=False, useBN=False, dropout=None,
       initializer="xavier", reuse=False, biasReg=False, biasInitializer="zeros"):
    """reference NNLayers.FC (NNLayers.py:98-115)."""
    inDim = inp.shape[1]
    temName = name if name is not None else "defaultParamName%d" % getParamId()
    W = getOrDefineParam(temName, [inDim, outDim], reg=reg, initializer=initializer, reuse=reuse)
    ret = (torch.nn.functional.dropout(inp, p=dropout) if dropout is not None else inp) @ W
    if useBias:
        ret = Bias(ret, name=name, reuse=reuse, reg=biasReg, initializer=biasInitializer)
    if activation is not None:
        ret = Activate(ret, activation)
    return ret


def Regularize(names=None, method="L2"):
    """reference NNLayers.Regularize (NNLayers.py:159-175)."""
    src = [getParam(n) for n in names] if names is not None else list(regParams.values())
    ret = 0
    for p in src:
        ret = ret + (p.abs().sum() if method == "L1" else p.square().sum())
    return ret
