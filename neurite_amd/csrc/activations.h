// Element-wise activations of the conv / LocallyConnected3D epilogues and of the stand-alone Activation layers, and their
// derivatives expressed through the OUTPUT y (the backward kernels keep y, not the pre-activation).
// The reference hands `activation` straight to Keras (neurite/tf/models.py:1346, 1429, 1507, 1588; layers.py:1101), so any of
// Keras' element-wise activation strings may arrive; definitions follow tf.keras.activations (TF 2.x):
//   elu (alpha 1), relu, sigmoid, tanh, softplus, softsign, selu, exponential, hard_sigmoid (0.2 x + 0.5 clipped),
//   leaky_relu (slope 0.2).  The channel softmax is not element-wise: the host runs it as its own kernel after a linear epilogue.
#pragma once

#include "nrt_common.h"

enum {
    ACT_NONE = 0, ACT_ELU = 1, ACT_RELU = 2, ACT_SIGMOID = 3, ACT_TANH = 4, ACT_SOFTPLUS = 5, ACT_SOFTSIGN = 6, ACT_SELU = 7,
    ACT_EXPONENTIAL = 8, ACT_HARD_SIGMOID = 9, ACT_LEAKY_RELU = 10, ACT_LAST = 10, ACT_MUL_B = 0x100
};

#define NRT_SELU_SCALE 1.05070098735548049342f
#define NRT_SELU_ALPHA 1.67326324235437728481f

// the activations fused into the conv / LocallyConnected3D epilogues (every other one runs as an element-wise pass over the layer
// output, nrt_add_act_affine_f32: inlining the whole table into 32 accumulators of a tile blew the kernels up and spilled)
__device__ __forceinline__ float nrt_activate_fused(float v, int act) {
    // exp on the hardware exponential (v_exp_f32 of v log2 e: relative error < 2e-6 for the arguments that matter) -- the libm expansion is
    // ~17 VALU instructions per output element in the epilogue of every MFMA tile
    if (act == ACT_ELU) return v > 0.0f ? v : (__builtin_amdgcn_exp2f(v * 1.44269504088896341f) - 1.0f);
    if (act == ACT_RELU) return fmaxf(v, 0.0f);
    return v;
}
#define ACT_LAST_FUSED ACT_RELU

__device__ __forceinline__ float nrt_activate(float v, int act) {
    switch (act) {
        case ACT_ELU: return v > 0.0f ? v : (expf(v) - 1.0f);                    // Keras elu: exp(x) - 1, not expm1
        case ACT_RELU: return fmaxf(v, 0.0f);
        case ACT_SIGMOID: return 1.0f / (1.0f + expf(-v));
        case ACT_TANH: return tanhf(v);
        case ACT_SOFTPLUS: return fmaxf(v, 0.0f) + log1pf(expf(-fabsf(v)));      // log(exp(x) + 1), the stable form TF uses
        case ACT_SOFTSIGN: return v / (1.0f + fabsf(v));
        case ACT_SELU: return NRT_SELU_SCALE * (v > 0.0f ? v : NRT_SELU_ALPHA * (expf(v) - 1.0f));
        case ACT_EXPONENTIAL: return expf(v);
        case ACT_HARD_SIGMOID: return fminf(fmaxf(0.2f * v + 0.5f, 0.0f), 1.0f);
        case ACT_LEAKY_RELU: return v > 0.0f ? v : 0.2f * v;
        default: return v;
    }
}

// d act / d pre-activation as a function of y = act(pre)
__device__ __forceinline__ float nrt_activate_slope(float y, int act) {
    switch (act) {
        case ACT_ELU: return y > 0.0f ? 1.0f : y + 1.0f;
        case ACT_RELU: return y > 0.0f ? 1.0f : 0.0f;
        case ACT_SIGMOID: return y * (1.0f - y);
        case ACT_TANH: return 1.0f - y * y;
        case ACT_SOFTPLUS: return 1.0f - expf(-y);                                // sigmoid(x) = 1 - exp(-softplus(x))
        case ACT_SOFTSIGN: { const float a = 1.0f - fabsf(y); return a * a; }     // 1 / (1 + |x|)^2 with |y| = |x| / (1 + |x|)
        case ACT_SELU: return y > 0.0f ? NRT_SELU_SCALE : y + NRT_SELU_SCALE * NRT_SELU_ALPHA;
        case ACT_EXPONENTIAL: return y;
        case ACT_HARD_SIGMOID: return (y > 0.0f && y < 1.0f) ? 0.2f : 0.0f;
        case ACT_LEAKY_RELU: return y > 0.0f ? 1.0f : 0.2f;
        default: return 1.0f;
    }
}
