"""`model.layers`, `model.inputs`, `Model(inputs=..., outputs=model.layers[i].output)`: CNN.ipynb cells 15 and 17 in the
notebook's own spelling.  The GPU test runs the cell against the oracle's named taps; the CPU tests hold the
layer list to the Sequential definitions the reference writes down and to what its bundled checkpoints name their layers."""
import os

import numpy as np
import pytest

from modulationdetectioncnn_amd import VTCNN2, Model, Topology, synthetic_frames, synthetic_weights
from oracle import oracle_np as O


def test_layer_list_is_the_notebooks(reference_dir):
    m = VTCNN2(Topology.deployed(3))
    # CNN.ipynb cell 15's comments: 0 Reshape, 1 Zero Padding, 2 Convolution + ReLU, 3 Flatten, 4 Dense + ReLU, 5 Softmax
    assert [l.class_name for l in m.layers] == ["Reshape", "ZeroPadding2D", "Conv2D", "Flatten", "Dense", "Activation", "Reshape"]
    assert repr(m.layers[2]) == "<Conv2D conv2d>" and m.layers[4].output_shape == (None, 3)
    assert sum(l.count_params() for l in m.layers) == 2334
    # the same rows the bundled checkpoint's own model_config lists
    from modulationdetectioncnn_amd.formats.h5mini import load_keras_h5
    ck = load_keras_h5(os.path.join(reference_dir, "3convmodrecnets_CNN2_0.5.wts.h5"))
    assert [l["class_name"] for l in ck.layer_configs() if l["class_name"] != "InputLayer"] == [l.class_name for l in m.layers]


def test_layer_weights_and_refusals():
    topo = Topology.cnnpy(10, 10, 5)
    w = synthetic_weights(topo, seed=4, bias_scale=0.1)
    m = VTCNN2(topo)
    with pytest.raises(RuntimeError):
        m.layers[2].get_weights()                                   # nothing loaded yet
    m.set_weights(w)
    assert [len(l.get_weights()) for l in m.layers] == [0, 0, 2, 0, 2, 2, 0, 0]
    for layer, (k, b) in zip([m.layers[2], m.layers[4], m.layers[5]], w):
        assert np.array_equal(layer.get_weights()[0], k) and np.array_equal(layer.get_weights()[1], b)
    with pytest.raises(ValueError, match="no tap"):
        Model(inputs=m.inputs, outputs=m.layers[1].output)          # a padding's output is never materialised
    with pytest.raises(ValueError, match="no tap"):
        v = VTCNN2(Topology.vtcnn2(11))
        Model(inputs=v.inputs, outputs=v.layers[2].output)          # VT-CNN2's conv1 feeds conv2 from registers
    with pytest.raises(ValueError):
        Model(inputs=VTCNN2(topo).inputs, outputs=m.layers[4].output)      # another model's inputs
    with pytest.raises(TypeError):
        Model(inputs=m.inputs, outputs="dense")
    assert Model(inputs=m.inputs, outputs=m.layers[4].output).output_shape == (None, 10)


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["deployed", "cnnpy", "vtcnn2"])
def test_cell_17_in_the_notebooks_spelling(kind):
    topo = {"deployed": Topology.deployed(3), "cnnpy": Topology.cnnpy(10, 10, 5), "vtcnn2": Topology.vtcnn2(11)}[kind]
    w = synthetic_weights(topo, seed=11, bias_scale=0.01)
    x = synthetic_frames(70, seed=5, sigma=0.05)
    ref = O.forward(kind, x, w, dtype=np.float64, **({"taps": True} if kind == "vtcnn2" else {}))
    model = VTCNN2(topo)
    model.set_weights(w)
    # layer index -> the oracle's name for that layer's output (CNN.ipynb cell 17: model2 = layers[4], model3 = [3], model4 = [2], model5 = [5])
    wanted = {"deployed": {4: "dense", 3: "flat", 2: "conv", 5: "probs"},
              "cnnpy": {2: "conv", 3: "flat", 4: "dense1", 5: "logits", 6: "probs"},
              "vtcnn2": {5: "conv2", 7: "flat", 8: "dense1", 10: "logits", 11: "probs"}}[kind]
    for idx, key in wanted.items():
        sub = Model(inputs=model.inputs, outputs=model.layers[idx].output)
        sub.compile(loss='categorical_crossentropy', optimizer='adam')
        out = sub.predict(np.array([x[2], ]))                          # the cell's own call shape: one frame
        assert out.shape == (1,) + model.layers[idx].output_shape[1:], (idx, out.shape)
        full = sub.predict(x)
        want = np.asarray(ref[key], np.float64).reshape(full.shape)
        scale = max(1.0, float(np.abs(want).max()))
        np.testing.assert_allclose(full, want, rtol=0, atol=2e-5 * scale, err_msg=f"{kind} layer {idx} ({key})")
        np.testing.assert_array_equal(out[0], full[2])                 # a frame's result does not depend on its batch
