/* CPU oracle, plain C — TEST INFRASTRUCTURE ONLY (never linked into the product library).
 *
 * An independent, loop-level restatement of the three arithmetic kernels of the hot path, used by
 * tests/test_oracle_c.py to cross-check the PyTorch-CPU oracle (oracle/decoder_ref.py) without sharing any
 * code or library with it:
 *   oracle_idwt_haar   IDWT(wave="haar", mode="zero")   reference closed form
 *                      /root/reference/KITTI/networks/decoders/depth_decoder.py:225-239
 *   oracle_dwt_haar    DWT(J=1, "haar", "reflect") on even sizes   /root/reference/NYUv2/train.py:258,289
 *   oracle_conv3x3     pad (zero/reflect/replicate) + 3x3 cross-correlation + bias, optional nearest x2 upsample
 *                      of the first C1 channels and concat of a skip tensor
 *                      /root/reference/KITTI/layers.py:146-161,233-236; depth_decoder.py:145-150;
 *                      /root/reference/NYUv2/networks/layers.py:11-32,57-67
 * Built by __graft_entry__.build():  gcc -O2 -shared -fPIC -o oracle/liboracle.so oracle/haar_conv_oracle.c -lm
 */
#include <stddef.h>

void oracle_idwt_haar(const float* yl, const float* yh, float* out, int N, int h, int w) {
    const size_t plane = (size_t)h * w;
    for (int n = 0; n < N; ++n)
        for (int i = 0; i < h; ++i)
            for (int j = 0; j < w; ++j) {
                const float a = yl[n * plane + (size_t)i * w + j];
                const float b = yh[(n * 3 + 0) * plane + (size_t)i * w + j];
                const float c = yh[(n * 3 + 1) * plane + (size_t)i * w + j];
                const float d = yh[(n * 3 + 2) * plane + (size_t)i * w + j];
                float* o = out + n * 4 * plane + (size_t)(2 * i) * (2 * w) + 2 * j;
                o[0] = (a + b + c + d) * 0.5f;
                o[1] = (a + b - c - d) * 0.5f;
                o[2 * w] = (a - b + c - d) * 0.5f;
                o[2 * w + 1] = (a - b - c + d) * 0.5f;
            }
}

void oracle_dwt_haar(const float* x, float* yl, float* yh, int N, int h, int w) {
    const size_t plane = (size_t)h * w;
    for (int n = 0; n < N; ++n)
        for (int i = 0; i < h; ++i)
            for (int j = 0; j < w; ++j) {
                const float* p = x + n * 4 * plane + (size_t)(2 * i) * (2 * w) + 2 * j;
                const float a = p[0], b = p[1], c = p[2 * w], d = p[2 * w + 1];
                yl[n * plane + (size_t)i * w + j] = (a + b + c + d) * 0.5f;
                yh[(n * 3 + 0) * plane + (size_t)i * w + j] = (a + b - c - d) * 0.5f;
                yh[(n * 3 + 1) * plane + (size_t)i * w + j] = (a - b + c - d) * 0.5f;
                yh[(n * 3 + 2) * plane + (size_t)i * w + j] = (a - b - c + d) * 0.5f;
            }
}

/* pad_mode: 0 zero, 1 reflect, 2 replicate.  Returns -1 when the tap reads zero. */
static int src_index(int g, int n, int pad_mode) {
    if (g >= 0 && g < n) return g;
    if (pad_mode == 1) return g < 0 ? -g : 2 * n - 2 - g;
    if (pad_mode == 2) return g < 0 ? 0 : n - 1;
    return -1;
}

void oracle_conv3x3(const float* x1, int C1, int up1, const float* x2, int C2, const float* wgt, const float* bias,
                    float* y, int B, int H, int W, int Cout, int pad_mode) {
    const int Cin = C1 + C2, H1 = H / up1, W1 = W / up1;
    for (int b = 0; b < B; ++b)
        for (int co = 0; co < Cout; ++co)
            for (int oy = 0; oy < H; ++oy)
                for (int ox = 0; ox < W; ++ox) {
                    double acc = bias ? bias[co] : 0.0;   /* double accumulation: an order-independent yardstick */
                    for (int ci = 0; ci < Cin; ++ci)
                        for (int ky = 0; ky < 3; ++ky) {
                            const int sy = src_index(oy + ky - 1, H, pad_mode);
                            if (sy < 0) continue;
                            for (int kx = 0; kx < 3; ++kx) {
                                const int sx = src_index(ox + kx - 1, W, pad_mode);
                                if (sx < 0) continue;
                                float v;
                                if (ci < C1)
                                    v = x1[(((size_t)b * C1 + ci) * H1 + sy / up1) * W1 + sx / up1];
                                else
                                    v = x2[(((size_t)b * C2 + (ci - C1)) * H + sy) * W + sx];
                                acc += (double)wgt[(((size_t)co * Cin + ci) * 3 + ky) * 3 + kx] * v;
                            }
                        }
                    y[(((size_t)b * Cout + co) * H + oy) * W + ox] = (float)acc;
                }
}
