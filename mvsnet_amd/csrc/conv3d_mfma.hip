// MFMA implicit-GEMM 3x3x3 convolution (placeholder until the tiled kernel lands).
#include "common.h"

int mvs_conv3d_mfma(const float*, const float*, const float*, const float*, const float*,
                    const float*, const float*, int, int, int, int, int, int, float*, double*,
                    hipStream_t) {
    return MVS_E_SHAPE;
}
int mvs_deconv3d_mfma(const float*, const float*, const float*, const float*, const float*,
                      const float*, const float*, int, int, int, int, int, float*, double*,
                      hipStream_t) {
    return MVS_E_SHAPE;
}
