// IoU / GIoU / DIoU / CIoU value and gradient for one box pair, fp32.
//
// Follows object_detection/yolov5/utils/metrics.py:216-240 (== object_detection/loss.py:138-166)
// operation by operation: h = y2-y1+eps, union = w1*h1 + w2*h2 - inter + eps, c2 = cw^2+ch^2+eps,
// c_area = cw*ch+eps, v = (4/pi^2)*(atan(w2/h2)-atan(w1/h1))^2 and alpha = v/(v-iou+(1+eps)) held
// constant for the gradient (the reference computes it under no_grad).  Built with
// -ffp-contract=off so no multiply-add is fused.
//
// one_plus_eps is (float)(1.0 + (double)eps): the reference forms 1 + eps in Python doubles first.
//
// Gradient conventions reproduce PyTorch autograd: min/max send the whole gradient to the
// selected operand and split it evenly on ties; clamp(0) passes gradient where its input >= 0.
#pragma once
#include "dh_common.h"

enum { DH_IOU = 0, DH_GIOU = 1, DH_DIOU = 2, DH_CIOU = 3 };

struct dh_corners {
    float x1, y1, x2, y2;
};

__device__ __forceinline__ dh_corners dh_from_xywh(float x, float y, float w, float h)
{
    dh_corners c;
    const float hw = w / 2, hh = h / 2;
    c.x1 = x - hw;
    c.x2 = x + hw;
    c.y1 = y - hh;
    c.y2 = y + hh;
    return c;
}

#define DH_FOUR_OVER_PI2 0.40528473456935109f   // (float)(4 / pi^2)

template <int MODE>
__device__ __forceinline__ float dh_iou_value(const dh_corners a, const dh_corners b, const float eps, const float one_plus_eps)
{
    const float iw = fmaxf(fminf(a.x2, b.x2) - fmaxf(a.x1, b.x1), 0.0f);
    const float ih = fmaxf(fminf(a.y2, b.y2) - fmaxf(a.y1, b.y1), 0.0f);
    const float inter = iw * ih;
    const float w1 = a.x2 - a.x1, h1 = a.y2 - a.y1 + eps;
    const float w2 = b.x2 - b.x1, h2 = b.y2 - b.y1 + eps;
    const float uni = w1 * h1 + w2 * h2 - inter + eps;
    const float iou = __fdiv_rn(inter, uni);
    if (MODE == DH_IOU) return iou;
    const float cw = fmaxf(a.x2, b.x2) - fminf(a.x1, b.x1);
    const float ch = fmaxf(a.y2, b.y2) - fminf(a.y1, b.y1);
    if (MODE == DH_GIOU) {
        const float c_area = cw * ch + eps;
        return iou - __fdiv_rn(c_area - uni, c_area);
    }
    const float c2 = cw * cw + ch * ch + eps;
    const float dx = b.x1 + b.x2 - a.x1 - a.x2;
    const float dy = b.y1 + b.y2 - a.y1 - a.y2;
    const float rho2 = (dx * dx + dy * dy) / 4;
    if (MODE == DH_DIOU) return iou - __fdiv_rn(rho2, c2);
    const float d = atanf(__fdiv_rn(w2, h2)) - atanf(__fdiv_rn(w1, h1));
    const float v = DH_FOUR_OVER_PI2 * (d * d);
    const float alpha = __fdiv_rn(v, v - iou + one_plus_eps);
    return iou - (__fdiv_rn(rho2, c2) + v * alpha);
}

// weight of operand p in d min(p,q)/dp and d max(p,q)/dp
__device__ __forceinline__ float dh_wmin(float p, float q) { return p < q ? 1.0f : (p == q ? 0.5f : 0.0f); }
__device__ __forceinline__ float dh_wmax(float p, float q) { return p > q ? 1.0f : (p == q ? 0.5f : 0.0f); }

// Value and gradient w.r.t. the corners of box a, for upstream gradient g.
template <int MODE>
__device__ __forceinline__ float dh_iou_value_grad(const dh_corners a, const dh_corners b, const float eps,
                                                   const float one_plus_eps, const float g, dh_corners& ga)
{
    const float iw_raw = fminf(a.x2, b.x2) - fmaxf(a.x1, b.x1);
    const float ih_raw = fminf(a.y2, b.y2) - fmaxf(a.y1, b.y1);
    const float iw = fmaxf(iw_raw, 0.0f), ih = fmaxf(ih_raw, 0.0f);
    const float inter = iw * ih;
    const float w1 = a.x2 - a.x1, h1 = a.y2 - a.y1 + eps;
    const float w2 = b.x2 - b.x1, h2 = b.y2 - b.y1 + eps;
    const float uni = w1 * h1 + w2 * h2 - inter + eps;
    const float iou = __fdiv_rn(inter, uni);
    float value = iou;

    float g_uni = 0.0f, g_w1 = 0.0f, g_h1 = 0.0f, g_cw = 0.0f, g_ch = 0.0f;
    float gx1 = 0.0f, gy1 = 0.0f, gx2 = 0.0f, gy2 = 0.0f;

    if (MODE != DH_IOU) {
        const float cw = fmaxf(a.x2, b.x2) - fminf(a.x1, b.x1);
        const float ch = fmaxf(a.y2, b.y2) - fminf(a.y1, b.y1);
        if (MODE == DH_GIOU) {
            const float c_area = cw * ch + eps;
            value = iou - __fdiv_rn(c_area - uni, c_area);
            // d/dc_area of -(c_area-uni)/c_area = -uni/c_area^2 ; d/duni = +1/c_area
            const float g_ca = -g * __fdiv_rn(uni, c_area * c_area);
            g_uni += __fdiv_rn(g, c_area);
            g_cw += g_ca * ch;
            g_ch += g_ca * cw;
        } else {
            const float c2 = cw * cw + ch * ch + eps;
            const float dx = b.x1 + b.x2 - a.x1 - a.x2;
            const float dy = b.y1 + b.y2 - a.y1 - a.y2;
            const float rho2 = (dx * dx + dy * dy) / 4;
            const float pen = __fdiv_rn(rho2, c2);
            // d(-rho2/c2): wrt rho2 = -1/c2, wrt c2 = rho2/c2^2
            const float g_rho2 = -__fdiv_rn(g, c2);
            const float g_c2 = g * __fdiv_rn(rho2, c2 * c2);
            g_cw += g_c2 * (2.0f * cw);
            g_ch += g_c2 * (2.0f * ch);
            const float hx = g_rho2 * 0.5f * dx, hy = g_rho2 * 0.5f * dy;   // d rho2/d a.x1 = -dx/2
            gx1 -= hx; gx2 -= hx; gy1 -= hy; gy2 -= hy;
            if (MODE == DH_CIOU) {
                const float q1 = __fdiv_rn(w1, h1);
                const float d = atanf(__fdiv_rn(w2, h2)) - atanf(q1);
                const float v = DH_FOUR_OVER_PI2 * (d * d);
                const float alpha = __fdiv_rn(v, v - iou + one_plus_eps);
                value = iou - (pen + v * alpha);
                // d(-alpha*v)/d atan(q1) = +alpha * (4/pi^2) * 2 d
                const float g_at1 = g * alpha * DH_FOUR_OVER_PI2 * (2.0f * d);
                const float g_q1 = __fdiv_rn(g_at1, 1.0f + q1 * q1);
                g_w1 += __fdiv_rn(g_q1, h1);
                g_h1 -= g_q1 * __fdiv_rn(q1, h1);
            } else {
                value = iou - pen;
            }
        }
    }
    // iou = inter/uni
    float g_inter = __fdiv_rn(g, uni);
    g_uni -= g * __fdiv_rn(iou, uni);
    // uni = w1*h1 + w2*h2 - inter + eps
    g_w1 += g_uni * h1;
    g_h1 += g_uni * w1;
    g_inter -= g_uni;
    // inter = clamp(iw_raw,0) * clamp(ih_raw,0)
    const float g_iw = (iw_raw >= 0.0f) ? g_inter * ih : 0.0f;
    const float g_ih = (ih_raw >= 0.0f) ? g_inter * iw : 0.0f;
    gx2 += g_iw * dh_wmin(a.x2, b.x2);
    gx1 -= g_iw * dh_wmax(a.x1, b.x1);
    gy2 += g_ih * dh_wmin(a.y2, b.y2);
    gy1 -= g_ih * dh_wmax(a.y1, b.y1);
    // cw = max(x2) - min(x1)
    gx2 += g_cw * dh_wmax(a.x2, b.x2);
    gx1 -= g_cw * dh_wmin(a.x1, b.x1);
    gy2 += g_ch * dh_wmax(a.y2, b.y2);
    gy1 -= g_ch * dh_wmin(a.y1, b.y1);
    // w1 = x2-x1, h1 = y2-y1+eps
    gx2 += g_w1; gx1 -= g_w1;
    gy2 += g_h1; gy1 -= g_h1;
    ga.x1 = gx1; ga.y1 = gy1; ga.x2 = gx2; ga.y2 = gy2;
    return value;
}

// corners-gradient -> (x,y,w,h)-gradient for x1 = x - w/2, x2 = x + w/2
__device__ __forceinline__ void dh_grad_to_xywh(const dh_corners g, float& gx, float& gy, float& gw, float& gh)
{
    gx = g.x1 + g.x2;
    gy = g.y1 + g.y2;
    gw = (g.x2 - g.x1) * 0.5f;
    gh = (g.y2 - g.y1) * 0.5f;
}
