/* szg/contraction.h — where a * b + c is evaluated with ONE rounding, by site class.
 *
 * The reference's shaders carry no `precise` qualifier and its committed SPIR-V no NoContraction decoration
 * (tests/test_spv_layout.py), so a Vulkan implementation may fuse any a * b + c; the kernels and the parity oracle fuse
 * explicitly, at the same places, and the compilers contract nothing by themselves (-ffp-contract=off). Round 2 had one
 * switch for all of those places; the result was up to 2.3e-3 away from a literal execution of the reference's SPIR-V
 * (tests/golden/spirv_vectors.npz), because some of the places feed the ill-conditioned part of the march
 * (1 - T(a) / T(b) over segments a few ulps of the planet radius long). Round 3 splits the places into classes, each a bit
 * of SZG_CONTRACT, measured one by one on MI355X against those vectors (profiles/r03_contraction_classes.md): a class is
 * fused in the product only if the product stays within 1e-4 relative and 1 UNORM16 step of every recorded value.
 *
 * Classes are defined by the SOURCE SITE in the reference's shaders, so that oracle and kernels agree on them:
 */
#ifndef SZG_CONTRACTION_H
#define SZG_CONTRACTION_H

/* OpDot / length / normalize / distance written in camera.comp, skyview_LUT.comp main, offscreen.vert/frag (view rays, the
 * sky-view LUT coordinates, the geometry term of camera.comp: everything in front of a march) */
#define SZG_C_DOT 0x001u
/* OpMatrixTimesVector / OpMatrixTimesMatrix rows: fma(m3, w, fma(m2, z, fma(m1, y, m0 * x))) */
#define SZG_C_MATVEC 0x002u
/* FMix: fma(b, w, a * (1 - w)) */
#define SZG_C_MIX 0x004u
/* the fixed-function LINEAR filter's weighted sum of four texels */
#define SZG_C_BILINEAR 0x008u
/* the fixed-function texel coordinate fma(s, W, -0.5) */
#define SZG_C_TEXCOORD 0x010u
/* textureCoordFromUnitRange, common.glinl:29-32: fma(x, 1 - 1/N, 0.5/N) */
#define SZG_C_LUTMAP 0x020u
/* transmittanceLUT_RMu_to_UV, common.glinl:40-66: d = max(fma(-r, mu, sqrt(fma(r*r, fma(mu, mu, -1), Ra^2))), 0) */
#define SZG_C_LUTDIST 0x040u
/* dot / length / normalize written in common.glinl: raySphereIntersection (:220-260), sampleTransmittanceLUT_Ray / _Segment
 * (:104-136), computeLuminanceScatteringIntegral (:364-424): radii and cosines of the atmosphere geometry */
#define SZG_C_ATMODOT 0x080u
/* the march's sample points origin - t * dir, common.glinl:386-387 */
#define SZG_C_POINT 0x100u
/* stepRadiusMu, common.glinl:316-334: fma(2 r mu, t, t*t) + r*r and fma(t, mu_step, r * mu_sun) */
#define SZG_C_STEP 0x200u
/* the march's two accumulations: fma(sM, pM, sR * pR) and luminance = fma(p * s * i, t, luminance), common.glinl:408-421 */
#define SZG_C_ACCUM 0x400u
/* length(position) of transmittance_LUT.comp:97 (the altitude of each of the 500 samples) */
#define SZG_C_TMAIN 0x800u

/* OpDot / normalize written in pbrFunctions.glinl:22-52 (half vectors of computeFresnel / specularBRDF and their cosines) */
#define SZG_C_PBRDOT 0x1000u
/* OpDot / normalize / distance written in lights.comp:65-108, :135 (light directions, N.L, view direction, falloff distances) */
#define SZG_C_LDOT 0x2000u

#define SZG_CONTRACT_ALL 0x3FFFu
#define SZG_CONTRACT_NONE 0x000u

/* The product's rule (round 3): the classes whose fusion, measured alone AND together on MI355X, leaves every one of the
 * 5 248 values recorded from the reference's SPIR-V within 1e-4 relative and 1 UNORM16 step (together: 6.6e-6 / 1 step on
 * camera.comp, 2.2e-7 on sky-view texels, transmittance texels bit-identical, lights 2.9e-6;
 * profiles/r03_contraction_classes.md). NOT fused: DOT (1.2e-3 alone: the view ray and the march length sit in front of the
 * ill-conditioned march), BILINEAR (2.8e-4), LUTDIST (2.1e-3), ATMODOT (1.8e-3 / 7 steps), POINT (3.2e-3), TMAIN (8.4e-4 through
 * the LUT everything else samples), and LUTMAP (6.7e-5 alone: inside the bar, but with a margin of 1.5 on these vectors
 * and 0.4 % of the frame time, it is left literal). */
#ifndef SZG_CONTRACT_DEFAULT
#define SZG_CONTRACT_DEFAULT (SZG_C_MATVEC | SZG_C_MIX | SZG_C_TEXCOORD | SZG_C_STEP | SZG_C_ACCUM | SZG_C_PBRDOT | SZG_C_LDOT)
#endif
#ifndef SZG_CONTRACT
#define SZG_CONTRACT SZG_CONTRACT_DEFAULT
#endif

/* a * b + c of site class `cls` */
#define SZG_CON(cls, a, b, c) (((SZG_CONTRACT) & (cls)) != 0u ? __builtin_fmaf((a), (b), (c)) : ((a) * (b) + (c)))

#endif
