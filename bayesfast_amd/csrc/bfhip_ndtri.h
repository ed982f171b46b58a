// bfhip_ndtri.h -- the normal quantile function on the device (scipy.special.ndtri = Cephes' ndtri.c; see bfhip_sit.hip).
#pragma once
__device__ inline double bf_horner(double t, const double *c, int n) {
#pragma clang fp contract(off)   // (as SciPy's build of Cephes: no fused multiply-adds)
    double v = c[0];
    for (int i = 1; i < n; ++i) v = v * t + c[i];
    return v;
}
__device__ inline double bf_ndtri(double p) {
#pragma clang fp contract(off)
    // numerators P*, denominators Q* (leading 1 written out), highest order first
    const double P0[5] = {-5.99633501014107895267E1, 9.80010754185999661536E1, -5.66762857469070293439E1, 1.39312609387279679503E1,
                          -1.23916583867381258016E0};
    const double Q0[9] = {1., 1.95448858338141759834E0, 4.67627912898881538453E0, 8.63602421390890590575E1, -2.25462687854119370527E2,
                          2.00260212380060660359E2, -8.20372256168333339912E1, 1.59056225126211695515E1, -1.18331621121330003142E0};
    const double P1[9] = {4.05544892305962419923E0, 3.15251094599893866154E1, 5.71628192246421288162E1, 4.40805073893200834700E1,
                          1.46849561928858024014E1, 2.18663306850790267539E0, -1.40256079171354495875E-1, -3.50424626827848203418E-2,
                          -8.57456785154685413611E-4};
    const double Q1[9] = {1., 1.57799883256466749731E1, 4.53907635128879210584E1, 4.13172038254672030440E1, 1.50425385692907503408E1,
                          2.50464946208309415979E0, -1.42182922854787788574E-1, -3.80806407691578277194E-2, -9.33259480895457427372E-4};
    const double P2[9] = {3.23774891776946035970E0, 6.91522889068984211695E0, 3.93881025292474443415E0, 1.33303460815807542389E0,
                          2.01485389549179081538E-1, 1.23716634817820021358E-2, 3.01581553508235416007E-4, 2.65806974686737550832E-6,
                          6.23974539184983293730E-9};
    const double Q2[9] = {1., 6.02427039364742014255E0, 3.67983563856160859403E0, 1.37702099489081330271E0, 2.16236993594496635890E-1,
                          1.34204006088543189037E-2, 3.28014464682127739104E-4, 2.89247864745380683936E-6, 6.79019408009981274425E-9};
    const double tail = 0.13533528323661269189;   // exp(-2)
    if (!(p >= 0. && p <= 1.)) return __builtin_nan("");
    if (p == 0.) return -__builtin_inf();
    if (p == 1.) return __builtin_inf();
    const bool upper = p > 1. - tail;
    double q = upper ? 1. - p : p;
    if (q > tail) {
        q -= 0.5;
        const double q2 = q * q;
        return (q + q * (q2 * bf_horner(q2, P0, 5) / bf_horner(q2, Q0, 9))) * 2.50662827463100050242;
    }
    const double z = sqrt(-2. * log(q)), r = 1. / z;
    const double corr = z < 8. ? r * bf_horner(r, P1, 9) / bf_horner(r, Q1, 9) : r * bf_horner(r, P2, 9) / bf_horner(r, Q2, 9);
    const double v = (z - log(z) / z) - corr;
    return upper ? v : -v;
}

