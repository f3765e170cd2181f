"""The reference's model DEFINITION in its own words (cnn.py:104-115; CNN.ipynb cell 6; the DeepSig notebook :229-243):

    from modulationdetectioncnn_amd.keras_mirror import models, callbacks, \\
        Reshape, Dense, Dropout, Activation, Flatten, Conv2D, Convolution2D, ZeroPadding2D
    model = models.Sequential()
    model.add(Reshape(in_shp+[1], input_shape=in_shp))
    model.add(ZeroPadding2D((0, 1)))
    model.add(Conv2D(3, (1, 2), activation='relu', padding='valid', input_shape=(1, 2, 128), kernel_initializer='glorot_uniform'))
    model.add(Flatten())
    model.add(Dense(len(classes), activation='relu', kernel_initializer='he_normal'))
    model.add(Activation('softmax'))
    model.add(Reshape([len(classes)]))
    model.compile(loss='categorical_crossentropy', optimizer='adam')
    model.build()
    model.summary()

The layer objects are DESCRIPTIONS.  `Sequential` checks the stack against the three definitions this library has kernels
for -- the deployed net (CNN.ipynb cell 6), cnn.py's literal net, the canonical VT-CNN2 -- and from then on IS a `VTCNN2`
(model.py) with freshly initialised weights (glorot_uniform convolutions, he_normal dense layers, zero biases, as the
definitions ask): every later call -- compile, fit, load_weights, predict, evaluate, layers -- is that class's.  Anything else
(another kernel size, 'same' padding, a pooling layer, a different activation) is refused by name: there is no general layer
interpreter behind this and no CPU fallback.  Dropout layers are accepted where the DeepSig definition has them; they are the
identity at inference, and training with them is `compile(dropout=rate)`."""
from __future__ import annotations

from typing import List, Optional, Sequence

import numpy as np

from . import callbacks  # noqa: F401   (`keras.callbacks`)
from .model import VTCNN2, Model
from .topology import Topology, synthetic_weights


class _LayerSpec:
    class_name = "?"

    def __init__(self, **config):
        self.config = config

    def keras_config(self) -> dict:
        return {"class_name": self.class_name, "config": dict(self.config)}

    def __repr__(self):
        return f"{self.class_name}({', '.join(f'{k}={v!r}' for k, v in self.config.items() if v is not None)})"


class Reshape(_LayerSpec):
    class_name = "Reshape"

    def __init__(self, target_shape, input_shape=None, name=None):
        super().__init__(target_shape=[int(d) for d in target_shape], input_shape=None if input_shape is None else [int(d) for d in input_shape],
                         name=name)


class ZeroPadding2D(_LayerSpec):
    class_name = "ZeroPadding2D"

    def __init__(self, padding=(1, 1), name=None):
        super().__init__(padding=tuple(int(p) for p in padding), name=name)


class Conv2D(_LayerSpec):
    class_name = "Conv2D"

    def __init__(self, filters, kernel_size, padding="valid", activation=None, input_shape=None, kernel_initializer="glorot_uniform",
                 data_format=None, strides=(1, 1), use_bias=True, name=None):
        super().__init__(filters=int(filters), kernel_size=[int(k) for k in kernel_size], padding=padding, activation=activation,
                         kernel_initializer=kernel_initializer, data_format=data_format, strides=tuple(strides), use_bias=bool(use_bias), name=name)


def Convolution2D(nb_filter, nb_row, nb_col, border_mode="valid", activation=None, name=None, init="glorot_uniform", **kw) -> Conv2D:
    """Keras 1's spelling, as the DeepSig notebook writes VT-CNN2 (:233-236) -- under its Theano ('th', channels_first) ordering."""
    return Conv2D(nb_filter, (nb_row, nb_col), padding=border_mode, activation=activation, kernel_initializer=init,
                  data_format="channels_first", name=name, **kw)


class Flatten(_LayerSpec):
    class_name = "Flatten"

    def __init__(self, name=None):
        super().__init__(name=name)


class Dense(_LayerSpec):
    class_name = "Dense"

    def __init__(self, units, activation=None, kernel_initializer="he_normal", init=None, use_bias=True, name=None):
        super().__init__(units=int(units), activation=activation, kernel_initializer=init or kernel_initializer, use_bias=bool(use_bias), name=name)


class Activation(_LayerSpec):
    class_name = "Activation"

    def __init__(self, activation, name=None):
        super().__init__(activation=activation, name=name)


class Dropout(_LayerSpec):
    class_name = "Dropout"

    def __init__(self, rate, name=None):
        super().__init__(rate=float(rate), name=name)


def recognise(specs: Sequence[_LayerSpec]) -> Topology:
    """The Topology a stack of layer descriptions defines, or a ValueError naming the first thing that is not one of the three
    definitions.  Beyond Topology.from_keras_config (which reads the shapes) this checks what a config of the REFERENCE's can
    leave unsaid but a hand-written stack can get wrong: paddings, activations, strides, biases, the softmax tail."""
    cfg = {"class_name": "Sequential", "config": {"name": "sequential", "layers": [s.keras_config() for s in specs]}}
    classes = [s.class_name for s in specs]
    body = [c for c in classes if c != "Dropout"]
    topo = Topology.from_keras_config(cfg)
    want = {"deployed": ["Reshape", "ZeroPadding2D", "Conv2D", "Flatten", "Dense", "Activation", "Reshape"],
            "cnnpy": ["Reshape", "ZeroPadding2D", "Conv2D", "Flatten", "Dense", "Dense", "Activation", "Reshape"],
            "vtcnn2": ["Reshape", "ZeroPadding2D", "Conv2D", "ZeroPadding2D", "Conv2D", "Flatten", "Dense", "Dense", "Activation", "Reshape"]}[topo.kind]
    if body != want:
        raise ValueError(f"layer order {classes} is not the {topo.kind} definition's {want} (Dropout layers aside)")
    if topo.kind != "vtcnn2" and "Dropout" in classes:
        raise ValueError("the deployed net and cnn.py's net have no Dropout layer (`dr` is never used there): train with compile(dropout=...) instead")
    convs = [s.config for s in specs if s.class_name == "Conv2D"]
    denses = [s.config for s in specs if s.class_name == "Dense"]
    pads = [s.config["padding"] for s in specs if s.class_name == "ZeroPadding2D"]
    reshapes = [s.config for s in specs if s.class_name == "Reshape"]
    for c in convs:
        if c["padding"] != "valid" or c["activation"] != "relu" or tuple(c["strides"]) != (1, 1) or not c["use_bias"]:
            raise ValueError(f"Conv2D must be padding='valid', activation='relu', stride 1, with bias; got {c}")
    if topo.kind == "vtcnn2":
        if [c["kernel_size"] for c in convs] != [[1, 3], [2, 3]] or any(c["data_format"] != "channels_first" for c in convs):
            raise ValueError("VT-CNN2 is Convolution2D(256, 1, 3) then Convolution2D(80, 2, 3) under channels_first ('th') ordering")
        if pads != [(0, 2), (0, 2)]:
            raise ValueError(f"VT-CNN2 pads by ZeroPadding2D((0, 2)) before each convolution; got {pads}")
    else:
        if pads != [(0, 1)] or any(c["data_format"] not in (None, "channels_last") for c in convs):
            raise ValueError(f"expected one ZeroPadding2D((0, 1)) and TensorFlow's channels_last ordering; got paddings {pads}")
    acts = [d["activation"] for d in denses]
    if acts != {"deployed": ["relu"], "cnnpy": ["relu", None], "vtcnn2": ["relu", None]}[topo.kind] or not all(d["use_bias"] for d in denses):
        raise ValueError(f"Dense activations {acts} are not the {topo.kind} definition's")
    if next(s for s in specs if s.class_name == "Activation").config["activation"] != "softmax":
        raise ValueError("the classifier ends in Activation('softmax')")
    if reshapes[0]["input_shape"] not in (None, [2, 128]) or reshapes[-1]["target_shape"] != [topo.classes]:
        raise ValueError("frames are (2, 128) and the last Reshape is [len(classes)]")
    return topo


class Sequential:
    """`models.Sequential()`: collects `add`ed layer descriptions; the first call of anything else turns it into the `VTCNN2`
    they describe (device, dtype: the constructor's; seed: the initialisers' -- Keras draws unseeded, None does the same)."""

    def __init__(self, layers: Optional[Sequence[_LayerSpec]] = None, device=None, dtype: str = "f32", seed: Optional[int] = None, name=None):
        self.__dict__["_specs"] = list(layers or [])
        self.__dict__["_args"] = dict(device=device, dtype=dtype)
        self.__dict__["_seed"] = seed
        self.__dict__["_model"] = None

    def add(self, layer: _LayerSpec) -> None:
        if self._model is not None:
            raise RuntimeError("the model is built: its definition is fixed")
        if not isinstance(layer, _LayerSpec):
            raise TypeError("add() takes the layer descriptions of modulationdetectioncnn_amd.keras_mirror")
        self._specs.append(layer)

    def built_model(self) -> VTCNN2:
        if self._model is None:
            topo = recognise(self._specs)
            m = VTCNN2(topo, **self._args)
            seed = int(np.random.default_rng().integers(0, 2 ** 31)) if self._seed is None else int(self._seed)
            m.set_weights(synthetic_weights(topo, seed=seed))
            self.__dict__["_model"] = m
        return self._model

    def build(self, input_shape=None) -> None:
        self.built_model()

    def __getattr__(self, name):            # only reached for what Sequential itself does not define: the VTCNN2's surface
        if name.startswith("__"):
            raise AttributeError(name)
        return getattr(self.built_model(), name)

    def __setattr__(self, name, value):
        setattr(self.built_model(), name, value)


class _Models:
    Sequential = Sequential
    Model = staticmethod(Model)


models = _Models()

__all__ = ["models", "callbacks", "Sequential", "Model", "Reshape", "ZeroPadding2D", "Conv2D", "Convolution2D", "Flatten", "Dense",
           "Activation", "Dropout", "recognise"]
