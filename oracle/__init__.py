"""
oracle/ -- CPU restatement of the reference (adalca/neurite) hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``neurite_amd/`` may import, call,
link or execute anything in this package.  The only permitted users are
``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` -- and there only as the checker / the timed CPU baseline,
never as the thing that is shipped.

Parity status
-------------
The reference cannot be imported as-is in this environment (TensorFlow, Keras,
voxelmorph and pystrum are absent), and it ships no tests
or golden vectors of its own.  The oracle is therefore pinned in two ways:

1. ``tests/golden/make_golden.py`` executes the reference's *own* Python source
   for ``interpn`` / ``resize`` / ``sub2ind2d`` / ``prod_n`` / ``meshgrid`` /
   ``batch_channel_flatten`` / ``Dice`` / the label-weighting step of
   ``CategoricalCrossentropy`` / ``LocallyConnected3D.local_conv`` straight from
   ``/root/reference`` on top of a small NumPy stand-in for the TensorFlow
   primitives those lines call (``tf.floor``, ``tf.gather`` ...).  The control
   flow, op order, index arithmetic and corner ordering in those vectors are the
   reference's, only the leaf primitives are restated.  The vectors are committed
   under ``tests/golden/`` and the oracle must reproduce them bit-for-bit.
2. Independent fp64 cross-checks (``scipy.ndimage.map_coordinates``, closed-form
   Dice/CCE, ``torch.nn.functional.conv3d`` on CPU).

What stays *unpinned* (no TF/Keras/voxelmorph source or binary available): the
leaf semantics of the TensorFlow primitives themselves (``tf.round`` half-to-even,
``tf.linspace`` rounding, Keras CCE epsilon/normalisation, Conv3D SAME padding,
ELU as exp(x)-1) and voxelmorph's ``SpatialTransformer`` (identity grid + shift),
which are restated from their published behaviour.  DESIGN.md repeats this.

Modules
-------
np_oracle.py   op-for-op NumPy fp32 restatement (slow, transparent)
oracle.c       the same algorithms in plain C (fast; used for full-size checks
               and as bench.py's cpu_baseline), built by oracle/build.py into
               oracle/_build/liboracle.so and wrapped by c_oracle.py
"""
