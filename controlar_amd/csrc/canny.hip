// canny.hip — Canny control extraction on the GPU: the step directly in front of the headline path,
// `cv2.Canny(img, 100, 200)` of condition/canny.py:6-14 as called at sample_t2i.py:123-125 (8-bit H x W x 3 photo -> H x W edge map
// in {0, 255}).  The arithmetic is OpenCV's (opencv-python==4.9.0.80, imgproc/src/canny.cpp; apertureSize 3, L1 gradient) —
// integer throughout, so the GPU result is bit-exact against the restatement in oracle/canny_oracle.py, which documents the
// algorithm and why it is "parity unpinned" here (cv2 is absent, no photo/edge pair exists in the reference tree).
//   canny_grad_nms  Sobel 3x3 (BORDER_REPLICATE) of the three channels from a 34x34 uint8 halo tile in LDS, per-pixel channel of
//                   maximum |dx|+|dy| -> mag tile (zero outside the image), fixed-point direction test (TG22), non-maximum
//                   suppression -> map: 0 weak candidate, 1 no edge, 2 strong seed.  One pass over the photo.
//   canny_hyst      hysteresis as a fixed point: a 32x32 tile (+1 halo) iterates "candidate with a strong 8-neighbour becomes strong"
//                   to convergence in LDS; launches repeat until no tile changed (propagation crosses one tile per launch).
//   canny_finish    map -> uint8 {0,255} and, optionally, the control tensor [3, H, W] = 2*(x/255 - 0.5) of sample_t2i.py:125,141.
#include "car_common.h"

#define CT 32
__global__ __launch_bounds__(256) void canny_grad_nms_kernel(const unsigned char* __restrict__ img, unsigned char* __restrict__ map, int H, int W, int low, int high) {
    __shared__ unsigned char px[3][CT + 4][CT + 4];       // photo tile with a 2-pixel halo (Sobel of the mag halo needs it)
    __shared__ short sdx[CT + 2][CT + 2], sdy[CT + 2][CT + 2];
    __shared__ int smag[CT + 2][CT + 2];
    const int b = blockIdx.z, x0 = blockIdx.x * CT, y0 = blockIdx.y * CT, tid = threadIdx.x;
    const unsigned char* im = img + (long)b * H * W * 3;
    for (int i = tid; i < (CT + 4) * (CT + 4); i += 256) {
        const int ly = i / (CT + 4), lx = i - ly * (CT + 4);
        int gy = y0 + ly - 2, gx = x0 + lx - 2;
        gy = gy < 0 ? 0 : (gy > H - 1 ? H - 1 : gy); gx = gx < 0 ? 0 : (gx > W - 1 ? W - 1 : gx);      // BORDER_REPLICATE
        const unsigned char* p = im + ((long)gy * W + gx) * 3;
        px[0][ly][lx] = p[0]; px[1][ly][lx] = p[1]; px[2][ly][lx] = p[2];
    }
    __syncthreads();
    for (int i = tid; i < (CT + 2) * (CT + 2); i += 256) {
        const int ly = i / (CT + 2), lx = i - ly * (CT + 2), gy = y0 + ly - 1, gx = x0 + lx - 1;
        int bdx = 0, bdy = 0, bm = 0;
        if (gy >= 0 && gy < H && gx >= 0 && gx < W) {          // magnitudes outside the image are zero
            // NOTE the replicate border is relative to the IMAGE edge: the px tile already holds clamped pixels
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const unsigned char (*q)[CT + 4] = px[c];
                const int cy = ly + 1, cx = lx + 1;
                const int dx = (q[cy - 1][cx + 1] + 2 * q[cy][cx + 1] + q[cy + 1][cx + 1]) - (q[cy - 1][cx - 1] + 2 * q[cy][cx - 1] + q[cy + 1][cx - 1]);
                const int dy = (q[cy + 1][cx - 1] + 2 * q[cy + 1][cx] + q[cy + 1][cx + 1]) - (q[cy - 1][cx - 1] + 2 * q[cy - 1][cx] + q[cy - 1][cx + 1]);
                const int m = abs(dx) + abs(dy);
                if (c == 0 || m > bm) { bdx = dx; bdy = dy; bm = m; }      // strict: the first channel wins ties
            }
        }
        sdx[ly][lx] = (short)bdx; sdy[ly][lx] = (short)bdy; smag[ly][lx] = bm;
    }
    __syncthreads();
    const int TG22 = 13573;                                   // round(tan(22.5 deg) * 2^15)
    for (int i = tid; i < CT * CT; i += 256) {
        const int ly = i / CT, lx = i - ly * CT, gy = y0 + ly, gx = x0 + lx;
        if (gy >= H || gx >= W) continue;
        const int cy = ly + 1, cx = lx + 1, m = smag[cy][cx];
        unsigned char out = 1;
        if (m > low) {
            const int xs = sdx[cy][cx], ys = sdy[cy][cx];
            const long x = abs(xs), y = (long)abs(ys) << 15, tg22x = x * TG22;
            bool keep;
            if (y < tg22x) keep = m > smag[cy][cx - 1] && m >= smag[cy][cx + 1];
            else {
                const long tg67x = tg22x + (x << 16);
                if (y > tg67x) keep = m > smag[cy - 1][cx] && m >= smag[cy + 1][cx];
                else { const int s = (xs ^ ys) < 0 ? -1 : 1; keep = m > smag[cy - 1][cx - s] && m > smag[cy + 1][cx + s]; }      // the two neighbours across the edge
            }
            if (keep) out = m > high ? 2 : 0;
        }
        map[((long)b * H + gy) * W + gx] = out;
    }
}

// One hysteresis sweep.  `prev` = the "something changed" flag of the previous sweep: once a sweep changes nothing the fixed point is reached and
// every later sweep of the batch exits at once (engine_encode.hip car_canny enqueues the sweeps in batches and looks at the LAST flag of a batch only).
__global__ __launch_bounds__(256) void canny_hyst_kernel(unsigned char* __restrict__ map, int H, int W, const int* __restrict__ prev, int* __restrict__ changed) {
    __shared__ unsigned char t[CT + 2][CT + 2];
    __shared__ int again, any;
    if (*prev == 0) return;
    const int b = blockIdx.z, x0 = blockIdx.x * CT, y0 = blockIdx.y * CT, tid = threadIdx.x;
    unsigned char* mp = map + (long)b * H * W;
    for (int i = tid; i < (CT + 2) * (CT + 2); i += 256) {
        const int ly = i / (CT + 2), lx = i - ly * (CT + 2), gy = y0 + ly - 1, gx = x0 + lx - 1;
        t[ly][lx] = (gy >= 0 && gy < H && gx >= 0 && gx < W) ? mp[(long)gy * W + gx] : (unsigned char)1;
    }
    if (tid == 0) any = 0;
    __syncthreads();
    for (int it = 0; it < 4 * CT; ++it) {
        if (tid == 0) again = 0;
        __syncthreads();
        for (int i = tid; i < CT * CT; i += 256) {
            const int cy = i / CT + 1, cx = i % CT + 1;
            if (t[cy][cx] == 0) {
                const bool nb = t[cy - 1][cx - 1] == 2 || t[cy - 1][cx] == 2 || t[cy - 1][cx + 1] == 2 || t[cy][cx - 1] == 2 || t[cy][cx + 1] == 2 ||
                                t[cy + 1][cx - 1] == 2 || t[cy + 1][cx] == 2 || t[cy + 1][cx + 1] == 2;
                if (nb) { t[cy][cx] = 2; again = 1; }            // monotone 0 -> 2: racing neighbours only speed it up
            }
        }
        __syncthreads();
        const int a = again;
        __syncthreads();
        if (!a) break;
        if (tid == 0) any = 1;
    }
    __syncthreads();
    if (any) {
        for (int i = tid; i < CT * CT; i += 256) {
            const int ly = i / CT, lx = i - ly * CT, gy = y0 + ly, gx = x0 + lx;
            if (gy < H && gx < W && t[ly + 1][lx + 1] == 2) mp[(long)gy * W + gx] = 2;
        }
        if (tid == 0) *changed = 1;
    }
}

template <typename T>
__global__ void canny_finish_kernel(const unsigned char* __restrict__ map, unsigned char* __restrict__ edges, void* __restrict__ control, int B, long HW) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x; const long st = (long)gridDim.x * blockDim.x;
    for (; i < B * HW; i += st) {
        const unsigned char e = map[i] == 2 ? 255 : 0;
        if (edges) edges[i] = e;
        if (control) {
            const long b = i / HW, p = i - b * HW;
            const float v = 2.0f * ((float)e / 255.0f - 0.5f);        // sample_t2i.py:141 on the 3x replicated map (:125)
            T* c = (T*)control + b * 3 * HW + p;
            ET<T>::st(c, v); ET<T>::st(c + HW, v); ET<T>::st(c + 2 * HW, v);
        }
    }
}

extern "C" void car_launch_canny_grad_nms(const unsigned char* img, unsigned char* map, int B, int H, int W, int low, int high, hipStream_t st) {
    hipLaunchKernelGGL(canny_grad_nms_kernel, dim3((W + CT - 1) / CT, (H + CT - 1) / CT, B), dim3(256), 0, st, img, map, H, W, low, high);
}
extern "C" void car_launch_canny_hyst(unsigned char* map, int B, int H, int W, const int* prev, int* changed, hipStream_t st) {
    hipLaunchKernelGGL(canny_hyst_kernel, dim3((W + CT - 1) / CT, (H + CT - 1) / CT, B), dim3(256), 0, st, map, H, W, prev, changed);
}
extern "C" void car_launch_canny_finish(int mode, const unsigned char* map, unsigned char* edges, void* control, int B, long HW, hipStream_t st) {
    long n = B * HW; int g = (int)((n + 255) / 256); if (g > 4096) g = 4096;
    if (mode == 1) hipLaunchKernelGGL(canny_finish_kernel<bf16_t>, dim3(g), dim3(256), 0, st, map, edges, control, B, HW);
    else hipLaunchKernelGGL(canny_finish_kernel<float>, dim3(g), dim3(256), 0, st, map, edges, control, B, HW);
}
