// grape_t18_coeffs.h -- coefficient sets of the five-product degree-18 polynomial evaluation (tools/t18_coeffs.py):
//     A2 = A A, A3 = A2 A, A6 = A3 A3
//     B1 = a1 A + a2 A2 + a3 A3,  B5 = e2 A2 + e3 A3 + e6 A6,  B4 = d0 I + d1 A + d2 A2 + d3 A3 + d6 A6
//     A9 = B1 B5 + B4,   p(A) = B2 + (B3 + A9) A9,   B3 = c0 I + c1 A + ... + c6 A6,  B2 = b0 I + b1 A + ... + b6 A6
// (scheme of Bader, Blanes, Casas 2019 for degree 18).  Order: a1, a2, a3, b1, b2, b3, b6, c0, c1, c2, c3, c6, d0, d1, d2, d3,
// d6, e2, e3, e6, b0.  Both sets have d0 = 0 and b0 = 1 (reached by continuation from the published solution, which has
// b0 = 0 and makes the identity as the product (c0 + d0) d0): the identity is ADDED at the end and the rounding error of A9 is
// not amplified into it -- max element error 2.0e-16 instead of 1.1e-15 at N = 64, rho = 1 (tools/t16_rounding.py).
//   * T18_*  : p = degree-18 Chebyshev truncation of exp on the segment i[-2, 2] -- for Hermitian generators
//              (A = -i dt H, spectrum on the imaginary axis): |p(-i lam) - exp(-i lam)| <= 1.6e-17 for |lam| <= T18_THETA = 2
//   * T18T_* : p = Taylor polynomial of degree 18 -- for general matrices; backward error below 2^-53 for
//              alpha <= T18T_THETA = 1.09, alpha = min(||A||_1, max(||A^2||_1^(1/2), ||A^3||_1^(1/3)))
//              (Al-Mohy, Higham 2009: the bound of the remainder series holds with alpha_p = max(d_p, d_{p+1}), p (p-1) <= 19)
#pragma once
#define T18_THETA 2.0
#define T18T_THETA 1.09
#define T18_A1 -0.10036558103014462001
#define T18_A2 -0.007456351650625886579
#define T18_A3 -0.00083091953191006175088
#define T18_B1 -0.14075572409909292308
#define T18_B2 1.0955465096236762141
#define T18_B3 0.29811198744557154804
#define T18_B6 -0.00057207257428924858549
#define T18_C0 -4.7094393853968811891
#define T18_C1 1.7157463766850012865
#define T18_C2 0.073686948027488562391
#define T18_C3 -0.0033650385206633560936
#define T18_C6 0.000033927981037541774044
#define T18_D0 0.0
#define T18_D1 -0.24222749901747747758
#define T18_D2 0.050668391204088569683
#define T18_D3 0.023404567895744140748
#define T18_D6 -0.000010355013205937047443
#define T18_E2 -0.13912895765004587534
#define T18_E3 -0.013910627366173824328
#define T18_E6 -0.000014649629174709440602
#define T18_B0 0.99999999999999999921
#define T18T_A1 -0.10036558103014462001
#define T18T_A2 -0.0080292464824115696008
#define T18T_A3 -0.00089213849804572995564
#define T18T_B1 0.24591022090110863764
#define T18T_B2 1.362667083208190483
#define T18T_B3 0.49892102569169427267
#define T18T_B6 -0.00064092743005853663879
#define T18T_C0 -11.148502971774368372
#define T18T_C1 1.6801581387890619718
#define T18T_C2 0.05717798464788655127
#define T18T_C3 -0.0069821012248805208429
#define T18T_C6 0.000033497501708607053831
#define T18T_D0 0.0
#define T18T_D1 -0.067640451907138190756
#define T18T_D2 0.067596130177045964608
#define T18T_D3 0.029555257042931552743
#define T18T_D6 -0.000013918025751606070112
#define T18T_E2 -0.092336461936711859281
#define T18T_E3 -0.01693649390020817172
#define T18T_E6 -0.00001400867981820361598
#define T18T_B0 1.0

// Four-product degree-16 evaluation (tools/t16_coeffs.py; scheme of Sastre 2018, the m = 15+ formulas):
//     A2 = A A,  y0 = A2 (c1 A2 + c2 A),  y1 = (y0 + c3 A2 + c4 A)(y0 + c5 A2) + c6 y0 + c7 A2
//     p(A) = (y1 + c8 A2 + c9 A)(y1 + c10 y0 + c11 A) + c12 y1 + c13 y0 + c14 A2 + c15 A + c16 I
//   * T16_C* : Hermitian generators with spectrum in [-T16_THETA, T16_THETA]: the coefficients 0..15 are those of the
//              degree-15 Chebyshev truncation of exp on the segment minus the part of the dependent top term b16 x^16
//              (b16 = c1^4 = 0.53 / 16!) that lower degrees can absorb; |p(-i lam) - exp(-i lam)| <= 8.9e-17 for
//              |lam| <= 1.36.  No scaling: cells beyond the bound take the degree-18 route.
#define T16_THETA 1.36
#define T16_C1 0.0003990485980387312666
#define T16_C2 0.0029228143276988285886
#define T16_C3 -0.0077470178547223424775
#define T16_C4 0.40523011494938120965
#define T16_C5 0.032514297214079020788
#define T16_C6 5.6985532981688705678
#define T16_C7 0.022467770221737013362
#define T16_C8 0.2375283509963317365
#define T16_C9 2.1834336740729687939
#define T16_C10 -5.7548650707099555138
#define T16_C11 -0.024893888233397960765
#define T16_C12 10.136883359859348227
#define T16_C13 -61.008736429296066557
#define T16_C14 0.32660098775353450156
#define T16_C15 0.9999999999999999025
#define T16_C16 0.99999999999999991128
