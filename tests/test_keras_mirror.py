"""keras_mirror: the reference's three model definitions typed as the reference types them (cnn.py:104-115, CNN.ipynb cell 6,
the DeepSig notebook :229-243) resolve to the topologies the kernels exist for; a stack that is not one of them is refused by
name.  The layer tables are the ones stored in the reference's notebooks (tests/test_summaries.py holds Topology.keras_layers
to those; here the definitions are held to Topology)."""
import numpy as np
import pytest

from modulationdetectioncnn_amd import Topology, synthetic_weights
from modulationdetectioncnn_amd.keras_mirror import (Activation, Conv2D, Convolution2D, Dense, Dropout, Flatten, Reshape, ZeroPadding2D,
                                                    callbacks, models, recognise)

in_shp = [2, 128]


def cnn_py_model(classes):            # cnn.py:104-112
    model = models.Sequential(seed=7)
    model.add(Reshape([1]+in_shp, input_shape=in_shp))
    model.add(ZeroPadding2D((0, 1)))
    model.add(Conv2D(10, (1, 2), padding='valid', input_shape=(1, 2, 128), activation="relu", kernel_initializer='glorot_uniform'))
    model.add(Flatten())
    model.add(Dense(10, activation='relu', kernel_initializer='he_normal'))
    model.add(Dense( len(classes), kernel_initializer='he_normal'))
    model.add(Activation('softmax'))
    model.add(Reshape([len(classes)]))
    return model


def notebook_model(classes, filters=3, **kw):      # CNN.ipynb cell 6
    model = models.Sequential(seed=7, **kw)
    model.add(Reshape(in_shp+[1], input_shape=in_shp))
    model.add(ZeroPadding2D((0, 1)))
    model.add(Conv2D(filters, (1, 2), activation='relu', padding='valid', input_shape=(1, 2, 128), kernel_initializer='glorot_uniform'))
    model.add(Flatten())
    model.add(Dense( len(classes),activation='relu', kernel_initializer='he_normal'))
    model.add(Activation('softmax'))
    model.add(Reshape([len(classes)]))
    return model


def deepsig_model(classes, dr=0.5):   # RML2016.10a_VTCNN2_example.ipynb:229-243 (Keras 1 spelling)
    model = models.Sequential(seed=7)
    model.add(Reshape([1]+in_shp, input_shape=in_shp))
    model.add(ZeroPadding2D((0, 2)))
    model.add(Convolution2D(256, 1, 3, border_mode='valid', activation="relu", name="conv1", init='glorot_uniform'))
    model.add(Dropout(dr))
    model.add(ZeroPadding2D((0, 2)))
    model.add(Convolution2D(80, 2, 3, border_mode="valid", activation="relu", name="conv2", init='glorot_uniform'))
    model.add(Dropout(dr))
    model.add(Flatten())
    model.add(Dense(256, activation='relu', init='he_normal', name="dense1"))
    model.add(Dropout(dr))
    model.add(Dense( len(classes), init='he_normal', name="dense2" ))
    model.add(Activation('softmax'))
    model.add(Reshape([len(classes)]))
    return model


def test_the_three_definitions_resolve(capsys):
    five, three, eleven = list("abcde"), list("abc"), list(range(11))
    m = cnn_py_model(five)
    m.compile(loss='categorical_crossentropy', optimizer='adam')       # cnn.py:113 (before build, as there)
    m.build()
    m.summary()
    assert m.topology == Topology.cnnpy(10, 10, 5) and "Total params: 2,935" in capsys.readouterr().out
    assert notebook_model(three).topology == Topology.deployed(3) and notebook_model(three, 10).topology == Topology.deployed(10)
    v = deepsig_model(eleven)
    assert v.topology == Topology.vtcnn2(11) and [l.class_name for l in v.layers].count("Dropout") == 3
    # freshly initialised as the definitions ask: glorot_uniform convolutions, he_normal dense layers, zero biases; seed = the constructor's
    for (k, b), (k2, b2) in zip(m.get_weights(), synthetic_weights(Topology.cnnpy(10, 10, 5), seed=7)):
        assert np.array_equal(k, k2) and np.array_equal(b, b2) and not b.any()
    specs = notebook_model(three)._specs
    a, b = models.Sequential(specs), models.Sequential(specs)
    assert not np.array_equal(a.get_weights()[1][0], b.get_weights()[1][0])      # seed=None draws afresh, as Keras does
    with pytest.raises(RuntimeError):
        m.add(Flatten())                                                           # built: the definition is fixed


@pytest.mark.parametrize("what,edit", [
    ("padding='same'", lambda L: L.__setitem__(2, Conv2D(3, (1, 2), activation='relu', padding='same'))),
    ("no relu on the conv", lambda L: L.__setitem__(2, Conv2D(3, (1, 2), padding='valid'))),
    ("kernel (1,3)", lambda L: L.__setitem__(2, Conv2D(3, (1, 3), activation='relu'))),
    ("pads by 2", lambda L: L.__setitem__(1, ZeroPadding2D((0, 2)))),
    ("dense without relu", lambda L: L.__setitem__(4, Dense(3))),
    ("sigmoid tail", lambda L: L.__setitem__(5, Activation('sigmoid'))),
    ("a Dropout the reference's net does not have", lambda L: L.insert(3, Dropout(0.5))),
    ("layers out of order", lambda L: L.reverse()),
    ("no flatten", lambda L: L.pop(3)),
    ("wrong class count in the last reshape", lambda L: L.__setitem__(6, Reshape([4]))),
    ("channels_first on the deployed net", lambda L: L.__setitem__(2, Conv2D(3, (1, 2), activation='relu', data_format='channels_first'))),
    ("frames of another length", lambda L: L.__setitem__(0, Reshape([2, 64, 1], input_shape=[2, 64]))),
])
def test_a_stack_that_is_not_one_of_them_is_refused(what, edit):
    L = list(notebook_model(list("abc"))._specs)
    edit(L)
    with pytest.raises(ValueError):
        recognise(L)
    with pytest.raises(ValueError):
        models.Sequential(L).build()


def test_vtcnn2_needs_theano_ordering():
    L = list(deepsig_model(list(range(11)))._specs)
    L[2] = Conv2D(256, (1, 3), activation='relu')                 # Keras 2 / TensorFlow ordering: a different net (cnn.py's accident)
    with pytest.raises(ValueError, match="channels_first"):
        recognise(L)
    with pytest.raises(TypeError):
        models.Sequential().add("Dense")


@pytest.mark.gpu
def test_cnn_py_from_definition_to_score(tmp_path):
    """cnn.py:104-153 typed as written there, on a separable stand-in for the data set."""
    from modulationdetectioncnn_amd.training import to_onehot
    from tests.test_training_gpu import _leveled
    x, lab = _leveled(3000, seed=5)
    X_train, Y_train, X_test, Y_test = x[:2000], to_onehot(lab[:2000], 3), x[2000:], to_onehot(lab[2000:], 3)
    classes = ['WBFM', 'AM-SSB', 'GFSK']
    model = notebook_model(classes, device=0)
    model.compile(loss='categorical_crossentropy', optimizer='adam', lr=0.01)
    model.build()
    model.summary()
    filepath = str(tmp_path / 'convmodrecnets_CNN2_0.5.wts.h5')
    history = model.fit(X_train, Y_train, batch_size=512, epochs=30, verbose=2, validation_data=(X_test, Y_test), seed=0,
                        callbacks=[callbacks.ModelCheckpoint(filepath, monitor='val_loss', verbose=0, save_best_only=True, mode='auto'),
                                   callbacks.EarlyStopping(monitor='val_loss', patience=5, verbose=0, mode='auto')])
    model.load_weights(filepath)
    score = model.evaluate(X_test, Y_test, verbose=0, batch_size=512)
    assert score < history.history['val_loss'][0] and abs(score - min(history.history['val_loss'])) <= 2e-5 * score
    test_Y_hat = model.predict(X_test, batch_size=512)
    assert test_Y_hat.shape == (1000, 3) and (test_Y_hat.argmax(1) == lab[2000:]).mean() > 0.5
    model2 = models.Model(inputs=model.inputs, outputs=model.layers[4].output)
    assert model2.predict(np.array([X_test[2], ])).shape == (1, 3)
