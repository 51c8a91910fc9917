"""MI355X-native drop-in for the `vo` package of saegsali/visual-odometry-project.

Same module layout, class names, keyword arguments and array conventions as the
reference (src/vo/...); the arithmetic of the per-frame front-end runs in
libvo_hip.so (HIP kernels for gfx950) through `vo._native`.  There is no CPU
fallback: without the library and a GPU the compute classes raise.
"""
