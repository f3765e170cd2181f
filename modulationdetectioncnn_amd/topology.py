"""Topologies of the VT-CNN2 family and their weights in the reference's (Keras) layouts.

Three families:
  deployed  CNN.ipynb cell 6 / model_config of the bundled .h5 (T1: F=3, T2: F=10)
  vtcnn2    examples-master/.../RML2016.10a_VTCNN2_example.ipynb:229-243 (T3, canonical)
  cnnpy     cnn.py:104-115 as TensorFlow actually builds it (T4: H=1, W=2, C=128)
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Sequence, Tuple

import numpy as np

CLASSES_3 = ["WBFM", "AM-SSB", "GFSK"]                 # CNN.ipynb cell 2 (index order of one-hot)
CLASSES_5 = ["BPSK", "GFSK", "QAM16", "QPSK", "WBFM"]  # cnn.py:47
CLASSES_11 = ["8PSK", "AM-DSB", "AM-SSB", "BPSK", "CPFSK", "GFSK", "PAM4", "QAM16", "QAM64", "QPSK", "WBFM"]  # CNN.ipynb cell 3


@dataclass(frozen=True)
class Topology:
    kind: str          # 'deployed' | 'vtcnn2' | 'cnnpy'
    filters: int
    hidden: int
    classes: int

    # -- constructors -------------------------------------------------------
    @staticmethod
    def deployed(filters: int = 3, classes: int = 3) -> "Topology":
        return Topology("deployed", int(filters), 0, int(classes))

    @staticmethod
    def vtcnn2(classes: int = 11) -> "Topology":
        return Topology("vtcnn2", 256, 256, int(classes))

    @staticmethod
    def cnnpy(filters: int = 10, hidden: int = 10, classes: int = 5) -> "Topology":
        return Topology("cnnpy", int(filters), int(hidden), int(classes))

    @staticmethod
    def from_keras_config(cfg: dict) -> "Topology":
        """Recognise a Keras ``model_config`` (the JSON attribute inside a .h5).  Anything else -- another model, or JSON that is
        not a Sequential config at all -- is a ValueError."""
        try:
            return Topology._from_keras_config(cfg)
        except (KeyError, TypeError, AttributeError, IndexError) as e:
            raise ValueError(f"not a Keras Sequential model_config ({type(e).__name__}: {e})") from e

    @staticmethod
    def _from_keras_config(cfg: dict) -> "Topology":
        layers = [(l["class_name"], l["config"]) for l in cfg["config"]["layers"] if l["class_name"] != "InputLayer"]
        names = [n for n, _ in layers]
        convs = [c for n, c in layers if n == "Conv2D"]
        denses = [c for n, c in layers if n == "Dense"]
        reshape = next((c for n, c in layers if n == "Reshape"), None)
        if not convs or not denses or reshape is None:
            raise ValueError(f"not a VT-CNN2-family model: layers {names}")
        tgt = list(reshape.get("target_shape", []))
        if tgt == [2, 128, 1] and len(convs) == 1 and len(denses) == 1 and list(convs[0]["kernel_size"]) == [1, 2]:
            if denses[0].get("activation") != "relu":
                raise ValueError("deployed net: Dense activation must be relu")
            return Topology.deployed(convs[0]["filters"], denses[0]["units"])
        if tgt == [1, 2, 128] and len(convs) == 1 and len(denses) == 2 and list(convs[0]["kernel_size"]) == [1, 2]:
            return Topology.cnnpy(convs[0]["filters"], denses[0]["units"], denses[1]["units"])
        if len(convs) == 2 and len(denses) == 2 and convs[0]["filters"] == 256 and convs[1]["filters"] == 80:
            return Topology.vtcnn2(denses[1]["units"])
        raise ValueError(f"unsupported topology: reshape {tgt}, convs {[c['filters'] for c in convs]}, denses {[d['units'] for d in denses]}")

    # -- shapes -------------------------------------------------------------
    @property
    def layer_names(self) -> List[str]:
        return {"deployed": ["conv", "dense"], "vtcnn2": ["conv1", "conv2", "dense1", "dense2"],
                "cnnpy": ["conv", "dense1", "dense2"]}[self.kind]

    @property
    def layer_shapes(self) -> List[Tuple[Tuple[int, ...], Tuple[int, ...]]]:
        """[(kernel_shape, bias_shape)] in the layout ``load_weights`` delivers."""
        F, D, C = self.filters, self.hidden, self.classes
        if self.kind == "deployed":
            return [((1, 2, 1, F), (F,)), ((258 * F, C), (C,))]                  # HWIO; (in,out)
        if self.kind == "vtcnn2":
            return [((256, 1, 1, 3), (256,)), ((80, 256, 2, 3), (80,)),             # OIHW (Keras-1 'th')
                    ((10560, 256), (256,)), ((256, C), (C,))]
        if self.kind == "cnnpy":
            return [((1, 2, 128, F), (F,)), ((3 * F, D), (D,)), ((D, C), (C,))]
        raise ValueError(self.kind)

    def keras_layers(self) -> List[Tuple[str, Tuple[int, ...], int]]:
        """The Sequential definition the reference writes down, layer by layer: (Keras-2 class name, output shape without
        the batch axis, parameter count) -- what ``model.summary()`` prints (cnn.py:115; CNN.ipynb cell 6; the DeepSig
        notebook :229-243, whose Keras-1 "Convolution2D" is Keras 2's Conv2D).  Derived from the topology alone: padding
        widths from the kernel widths, conv outputs from 'valid' convolution over the padded input, channels_last for the
        deployed nets and cnn.py's literal model, channels_first (Keras-1 / Theano ordering) for the canonical VT-CNN2.
        tests/test_summaries.py holds every row against the tables stored in the reference's notebooks."""
        F, D, C = self.filters, self.hidden, self.classes
        if self.kind == "deployed":       # Reshape([2,128,1]) . ZeroPadding2D((0,1)) . Conv2D(F,(1,2)) . Flatten . Dense(C,relu) . softmax . Reshape([C])
            return [("Reshape", (2, 128, 1), 0), ("ZeroPadding2D", (2, 130, 1), 0), ("Conv2D", (2, 130 - 2 + 1, F), (1 * 2 * 1 + 1) * F),
                    ("Flatten", (2 * 129 * F,), 0), ("Dense", (C,), (2 * 129 * F + 1) * C), ("Activation", (C,), 0), ("Reshape", (C,), 0)]
        if self.kind == "cnnpy":          # cnn.py:104-112 as TensorFlow reads it: (H, W, C) = (1, 2, 128)
            return [("Reshape", (1, 2, 128), 0), ("ZeroPadding2D", (1, 4, 128), 0), ("Conv2D", (1, 4 - 2 + 1, F), (1 * 2 * 128 + 1) * F),
                    ("Flatten", (3 * F,), 0), ("Dense", (D,), (3 * F + 1) * D), ("Dense", (C,), (D + 1) * C), ("Activation", (C,), 0),
                    ("Reshape", (C,), 0)]
        if self.kind == "vtcnn2":         # channels_first: (C, H, W)
            return [("Reshape", (1, 2, 128), 0), ("ZeroPadding2D", (1, 2, 132), 0), ("Conv2D", (256, 2, 132 - 3 + 1), (1 * 1 * 3 + 1) * 256),
                    ("Dropout", (256, 2, 130), 0), ("ZeroPadding2D", (256, 2, 134), 0), ("Conv2D", (80, 2 - 2 + 1, 134 - 3 + 1), (256 * 2 * 3 + 1) * 80),
                    ("Dropout", (80, 1, 132), 0), ("Flatten", (80 * 132,), 0), ("Dense", (256,), (10560 + 1) * 256), ("Dropout", (256,), 0),
                    ("Dense", (C,), (256 + 1) * C), ("Activation", (C,), 0), ("Reshape", (C,), 0)]
        raise ValueError(self.kind)

    def summary(self) -> str:
        """A ``model.summary()``-style table (cnn.py:115) of keras_layers()."""
        rows = self.keras_layers()
        lines = ["Layer (type)                 Output Shape              Param #", "=" * 65]
        for name, shape, params in rows:
            lines.append(f"{name:<28} {str((None,) + tuple(shape)):<25} {params}")
        lines += ["=" * 65, f"Total params: {sum(p for _, _, p in rows):,}"]
        return "\n".join(lines)

    @property
    def flatten_order(self) -> str:
        return "channels_first" if self.kind == "vtcnn2" else "channels_last"

    @property
    def flops_per_frame(self) -> int:
        """2 FLOP per MAC; bias/ReLU/softmax ignored; padding MACs included."""
        F, D, C = self.filters, self.hidden, self.classes
        if self.kind == "deployed":
            return 2 * (258 * F * 2 + 258 * F * C)
        if self.kind == "vtcnn2":
            return 2 * (256 * 2 * 130 * 3 + 80 * 132 * 256 * 6 + 10560 * 256 + 256 * C)
        return 2 * (3 * F * 256 + 3 * F * D + D * C)

    @property
    def conv_flops_per_frame(self) -> int:
        if self.kind != "vtcnn2":
            raise ValueError("conv-only count is defined for vtcnn2")
        return 2 * (256 * 2 * 130 * 3 + 80 * 132 * 256 * 6)

    @property
    def io_bytes_per_frame(self) -> int:
        """Algorithmic HBM bytes: f32 frame in + f32 probabilities out."""
        return 2 * 128 * 4 + 4 * self.classes

    def class_names(self) -> List[str]:
        return {3: CLASSES_3, 5: CLASSES_5, 11: CLASSES_11}.get(self.classes, [str(i) for i in range(self.classes)])


def glorot_uniform(rng: np.random.Generator, shape: Sequence[int], fan_in: int, fan_out: int) -> np.ndarray:
    lim = np.sqrt(6.0 / (fan_in + fan_out))
    return rng.uniform(-lim, lim, size=shape).astype(np.float32)


def he_normal(rng: np.random.Generator, shape: Sequence[int], fan_in: int) -> np.ndarray:
    return (rng.standard_normal(size=shape) * np.sqrt(2.0 / fan_in)).astype(np.float32)


def synthetic_weights(topo: Topology, seed: int = 2016, bias_scale: float = 0.0) -> List[Tuple[np.ndarray, np.ndarray]]:
    """Random-init weights with the reference's initialisers: glorot_uniform convs, he_normal
    denses, zero biases (cnn.py:108-111; RML2016.10a_VTCNN2_example.ipynb:233-241).
    ``bias_scale`` > 0 draws N(0, bias_scale) biases instead, so tests exercise the bias paths."""
    rng = np.random.default_rng(seed)
    out = []
    for (ks, bs) in topo.layer_shapes:
        if len(ks) == 4:
            if topo.kind == "vtcnn2":       # OIHW
                o, i, kh, kw = ks
            else:                           # HWIO
                kh, kw, i, o = ks
            k = glorot_uniform(rng, ks, i * kh * kw, o * kh * kw)
        else:
            k = he_normal(rng, ks, ks[0])
        b = (rng.standard_normal(bs) * bias_scale).astype(np.float32) if bias_scale > 0 else np.zeros(bs, np.float32)
        out.append((k, b))
    return out


def synthetic_frames(n: int, seed: int = 2016, sigma: float = 5e-3, device=None):
    """X ~ N(0, sigma) float32 (n,2,128): the bundled frames' scale.
    numpy on host, or a torch tensor generated directly on `device`."""
    if device is None:
        rng = np.random.default_rng(seed)
        return (rng.standard_normal((n, 2, 128)) * sigma).astype(np.float32)
    import torch
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    return torch.randn((n, 2, 128), generator=g, device=device, dtype=torch.float32) * sigma


# ----------------------------------------------------------------------------------------------------------------------
# Keras-side description of a topology: what a full-model .h5 carries in its `model_config` / `training_config`
# attributes (formats/h5mini.write_keras_h5).  Key order and values follow the bundled files' own JSON (Keras 2.4.0).
# ----------------------------------------------------------------------------------------------------------------------
def _keras_roles(kind: str) -> List[str]:
    """The Sequential's layers by role, in order (cnn.py:104-112; CNN.ipynb cell 6; the DeepSig notebook :229-243)."""
    if kind == "deployed":
        return ["reshape", "pad", "conv", "flatten", "dense", "activation", "reshape"]
    if kind == "cnnpy":
        return ["reshape", "pad", "conv", "flatten", "dense", "dense", "activation", "reshape"]
    if kind == "vtcnn2":
        return ["reshape", "pad", "conv", "dropout", "pad", "conv", "dropout", "flatten", "dense", "dropout", "dense", "activation", "reshape"]
    raise ValueError(kind)


_KERAS_BASENAME = {"reshape": "reshape", "pad": "zero_padding2d", "conv": "conv2d", "dropout": "dropout", "flatten": "flatten",
                   "dense": "dense", "activation": "activation"}


def _topology_keras_layer_names(self, override=None) -> List[Tuple[str, str]]:
    """[(role, Keras layer name)]: the names a fresh Keras session generates (reshape, zero_padding2d, conv2d, flatten,
    dense, dense_1, activation, reshape_1), or `override` -- a list of names, one per layer, e.g. the bundled files'
    reshape_6 / zero_padding2d_3 / ... -- to reproduce an existing file's."""
    roles = _keras_roles(self.kind)
    if override is not None:
        override = list(override)
        if len(override) != len(roles):
            raise ValueError(f"need {len(roles)} layer names, got {len(override)}")
        return list(zip(roles, override))
    seen: dict = {}
    out = []
    for r in roles:
        k = seen.get(r, 0)
        seen[r] = k + 1
        out.append((r, _KERAS_BASENAME[r] + (f"_{k}" if k else "")))
    return out


Topology.keras_layer_names = _topology_keras_layer_names


def _init(name: str) -> dict:
    return {"class_name": name, "config": {"seed": None}} if name != "Zeros" else {"class_name": "Zeros", "config": {}}


def _conv_cfg(name, filters, ksize, data_format, first):
    cfg = {"name": name, "trainable": True}
    if first is not None:
        cfg["batch_input_shape"] = first
    cfg.update({"dtype": "float32", "filters": filters, "kernel_size": list(ksize), "strides": [1, 1], "padding": "valid",
                "data_format": data_format, "dilation_rate": [1, 1], "groups": 1, "activation": "relu", "use_bias": True,
                "kernel_initializer": _init("GlorotUniform"), "bias_initializer": _init("Zeros"), "kernel_regularizer": None,
                "bias_regularizer": None, "activity_regularizer": None, "kernel_constraint": None, "bias_constraint": None})
    return {"class_name": "Conv2D", "config": cfg}


def _dense_cfg(name, units, activation):
    return {"class_name": "Dense", "config": {
        "name": name, "trainable": True, "dtype": "float32", "units": units, "activation": activation, "use_bias": True,
        "kernel_initializer": _init("HeNormal"), "bias_initializer": _init("Zeros"), "kernel_regularizer": None,
        "bias_regularizer": None, "activity_regularizer": None, "kernel_constraint": None, "bias_constraint": None}}


def keras_model_config(topo: Topology, names: Sequence[Tuple[str, str]], model_name: str = "sequential") -> dict:
    """The `model_config` JSON of a Keras 2.4 save of this Sequential.  For the deployed nets it reproduces the bundled
    files' attribute byte for byte (tests/test_h5_writer.py) -- including `batch_input_shape: [null, 1, 2, 128]` on the
    Conv2D, the reference's stray `input_shape=(1, 2, 128)` argument (CNN.ipynb cell 6; cnn.py:108)."""
    F, D, C = topo.filters, topo.hidden, topo.classes
    cf = topo.kind == "vtcnn2"
    fmt = "channels_first" if cf else "channels_last"
    target = {"deployed": [2, 128, 1], "cnnpy": [1, 2, 128], "vtcnn2": [1, 2, 128]}[topo.kind]
    pad = 2 if cf else 1
    convs = {"deployed": [(F, (1, 2))], "cnnpy": [(F, (1, 2))], "vtcnn2": [(256, (1, 3)), (80, (2, 3))]}[topo.kind]
    denses = {"deployed": [(C, "relu")], "cnnpy": [(D, "relu"), (C, "linear")], "vtcnn2": [(256, "relu"), (C, "linear")]}[topo.kind]
    layers = [{"class_name": "InputLayer", "config": {"batch_input_shape": [None, 2, 128], "dtype": "float32", "sparse": False,
                                                       "ragged": False, "name": names[0][1] + "_input"}}]
    ci = di = ri = 0
    for role, name in names:
        if role == "reshape":
            cfg = {"name": name, "trainable": True}
            if ri == 0:
                cfg["batch_input_shape"] = [None, 2, 128]
            cfg.update({"dtype": "float32", "target_shape": target if ri == 0 else [C]})
            layers.append({"class_name": "Reshape", "config": cfg})
            ri += 1
        elif role == "pad":
            layers.append({"class_name": "ZeroPadding2D", "config": {"name": name, "trainable": True, "dtype": "float32",
                                                                     "padding": [[0, 0], [pad, pad]], "data_format": fmt}})
        elif role == "conv":
            f, ks = convs[ci]
            layers.append(_conv_cfg(name, f, ks, fmt, [None, 1, 2, 128] if ci == 0 and not cf else None))
            ci += 1
        elif role == "dropout":
            layers.append({"class_name": "Dropout", "config": {"name": name, "trainable": True, "dtype": "float32", "rate": 0.5,
                                                               "noise_shape": None, "seed": None}})
        elif role == "flatten":
            layers.append({"class_name": "Flatten", "config": {"name": name, "trainable": True, "dtype": "float32", "data_format": fmt}})
        elif role == "dense":
            u, act = denses[di]
            layers.append(_dense_cfg(name, u, act))
            di += 1
        elif role == "activation":
            layers.append({"class_name": "Activation", "config": {"name": name, "trainable": True, "dtype": "float32", "activation": "softmax"}})
    return {"class_name": "Sequential", "config": {"name": model_name, "layers": layers}}


def keras_training_config(adam: dict) -> dict:
    """`training_config` of model.compile(loss='categorical_crossentropy', optimizer='adam') (cnn.py:113): Keras stores the
    float32 hyper-parameters as the doubles they round-trip to; epsilon is a Python float."""
    f = lambda v: float(np.float32(v))
    return {"loss": "categorical_crossentropy", "metrics": None, "weighted_metrics": None, "loss_weights": None,
            "optimizer_config": {"class_name": "Adam", "config": {
                "name": "Adam", "learning_rate": f(adam.get("lr", 1e-3)), "decay": 0.0, "beta_1": f(adam.get("beta1", 0.9)),
                "beta_2": f(adam.get("beta2", 0.999)), "epsilon": float(adam.get("eps", 1e-7)), "amsgrad": False}}}
