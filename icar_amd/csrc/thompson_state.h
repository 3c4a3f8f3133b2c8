// icar_amd/csrc/thompson_state.h -- constants and table state of the Thompson scheme (rows M3/M4):
// the module-level PARAMETERs and SAVE variables of src/physics/mp_thompson.f90:50-330, held per
// device context instead of in module state.  Host copy `ThState` owns host tables; the device copy
// has the same layout with device pointers.
#pragma once
#include <stddef.h>

#define NBINS 100
#define NTB_C 37
#define NTB_I 64
#define NTB_R 37
#define NTB_S 28
#define NTB_G 28
#define NTB_G1 28
#define NTB_R1 37
#define NTB_I1 55
#define NTB_T 9
#define TH_P10_N 84            /* 10**n, n = -TH_P10_OFF .. TH_P10_N-TH_P10_OFF-1 (= -45 .. 38) */
#define TH_P10_OFF 45

#define TH_T_0 273.15f
#define TH_PI2 3.1415926536f
#define TH_rho_w 1000.0f
#define TH_rho_s 100.0f
#define TH_rho_i 890.0f
#define TH_mu_g 0.0f
#define TH_mu_i 0.0f
#define TH_mu_s 0.6357f
#define TH_Kap0 490.6f
#define TH_Kap1 17.46f
#define TH_Lam0 20.78f
#define TH_Lam1 3.29f
#define TH_gonv_min 1.E4f
#define TH_gonv_max 3.E6f
#define TH_am_r (TH_PI2 * TH_rho_w / 6.0f)
#define TH_bm_r 3.0f
#define TH_bm_s 2.0f
#define TH_bm_g 3.0f
#define TH_am_i (TH_PI2 * TH_rho_i / 6.0f)
#define TH_bm_i 3.0f
#define TH_av_r 4854.0f
#define TH_bv_r 1.0f
#define TH_fv_r 195.0f
#define TH_bv_i 1.0f
#define TH_C_cube 0.5f
#define TH_R1 1.E-12f
#define TH_R2 1.E-6f
#define TH_eps 1.E-15f
#define TH_ATO 0.304f
#define TH_rho_not (101325.0f / (287.05f * 298.0f))
#define TH_Sc 0.632f
#define TH_HGFR 235.16f
#define TH_Rv 461.5f
#define TH_oRv (1.f / TH_Rv)
#define TH_RR2 287.04f
#define TH_Cp2 1004.0f
#define TH_lsub 2.834E6f
#define TH_lvap0 2.5E6f
#define TH_lfus (TH_lsub - TH_lvap0)
#define TH_olfus (1.f / TH_lfus)
#define TH_xm0i 1.E-12f
#define TH_D0c 1.E-6f
#define TH_D0r 50.E-6f
#define TH_D0s 200.E-6f
#define TH_D0g 250.E-6f

struct ThState {
    int initialized;
    /* mp_options (:327-330) */
    float Nt_c, TNO, am_s, rho_g, av_s, bv_s, fv_s, av_g, bv_g, av_i, Ef_si, Ef_rs, Ef_rg, Ef_ri;
    float C_cubes, C_sqrd, mu_r, t_adjust, am_g;
    int Ef_rw_l, Ef_sw_l;
    float mu_c, Sc3, D0i, xm0s, xm0g;
    float cce[3], ccg[3], ocg1, ocg2;
    float cie[7], cig[7], oig1, oig2, obmi;
    float cre[13], crg[13], ore1, org1, org2, org3, obmr;
    float cse[18], csg[18], oams, obms, ocms;
    float cge[12], cgg[12], oge1, ogg1, ogg2, ogg3, oamg, obmg, ocmg;
    float t1_qr_qc, t1_qr_qi, t2_qr_qi, t1_qg_qc, t1_qs_qc, t1_qs_qi, t1_qr_ev, t2_qr_ev;
    float t1_qs_sd, t2_qs_sd, t1_qg_sd, t2_qg_sd, t1_qs_me, t2_qs_me, t1_qg_me, t2_qg_me;
    int nic2, nii2, nii3, nir2, nir3, nis2, nig2, nig3;
    float r_c[NTB_C], r_i[NTB_I], r_r[NTB_R], r_g[NTB_G], r_s[NTB_S], N0r_exp[NTB_R1], N0g_exp[NTB_G1], Nt_i[NTB_I1];
    double Dc[NBINS], dtc[NBINS], Di[NBINS], dti[NBINS], Dr[NBINS], dtr[NBINS], Ds[NBINS], dts[NBINS], Dg[NBINS], dtg[NBINS];
    /* lookup tables, Fortran order */
    double *tcg_racg, *tmr_racg, *tcr_gacr, *tmg_gacr, *tnr_racg, *tnr_gacr;                  /* (28,28,37,37) */
    double *tcs_racs1, *tmr_racs1, *tcs_racs2, *tmr_racs2, *tcr_sacr1, *tms_sacr1, *tcr_sacr2, *tms_sacr2,
           *tnr_racs1, *tnr_racs2, *tnr_sacr1, *tnr_sacr2;                                   /* (28,9,37,37) */
    double *tpi_qcfz, *tni_qcfz;                                                              /* (37,45) */
    double *tpi_qrfz, *tpg_qrfz, *tni_qrfz, *tnr_qrfz;                                        /* (37,37,45) */
    double *tps_iaus, *tni_iaus, *tpi_ide;                                                    /* (64,55) */
    double *t_Efrw, *t_Efsw;                                                                  /* (100,100) */
    float sa[10], sb[10], Tc[NTB_T];
    /* graupel intercept of a level without graupel above 5e-5 and without supercooled rain (mp_thompson.f90:1456-1466 with
     * rg <= 5e-5, xslw1 = 0.01): a constant, evaluated once on the device by the very function the levels use */
    double N0_exp_default;
    /* more values a level would otherwise recompute from constants, produced by the level code's own functions
     * (k_thompson_constants): (cgg(3)*ogg2*ogg1)**obmg (:1464, :2387), (ccg(3)*ocg2)**obmr (:1518),
     * log(Dr(nbr)/Dr(1)) and log(Ds(nbs)/Ds(1)) (:1533, :1703), and 10.**n for the decade indices (:1562-1627) */
    float pw_cgg_obmg, pw_ccg_obmr;
    double log_Dr_span, log_Ds_span;
    float p10[TH_P10_N];
};
