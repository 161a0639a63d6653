/*
 * C restatement of the reference's CPU algorithm for the hot path (OpenMP), used as
 *   (i)  a second, independently written checker beside oracle/ldw_oracle.py, and
 *   (ii) the `cpu_baseline` ("port") of bench.py, timed on the GPU box's host cores.
 *
 * TEST INFRASTRUCTURE ONLY — the product (ldweaver_amd/) never links, loads or calls this file.
 * PARITY UNPINNED — see the header of oracle/ldw_oracle.py: the reference (R + Rcpp + MatrixExtra)
 * cannot be built or run in this image and its tests hold no golden values for this path.
 *
 * Citations are file:line in the reference checkout.
 *
 * build: make -C oracle      (gcc -O3 -fopenmp -shared -fPIC)
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* src/computeMI.cpp:11-21 — same expression, same association order, linear index, OpenMP parallel for */
void orc_fast_hadamard(double *MI, const double *den, const double *uq, const double *pxy, const double *pxpy,
                       const double *RXY, const double *pXrX, const double *pYrY, int64_t n, int ncores) {
#pragma omp parallel for num_threads(ncores)
    for (int64_t c = 0; c < n; c++)
        MI[c] += uq[c] * pxy[c] / den[c] * log(pxy[c] / (pxpy[c] + RXY[c] + pXrX[c] + pYrY[c]) * den[c]);
}

/* src/ACGTN2num_parallel.cpp:10-43 */
void orc_acgtn2num(double *nv, const char *ref, int64_t L, int ncores) {
#pragma omp parallel for num_threads(ncores)
    for (int64_t c = 0; c < L; c++) {
        char cc = ref[c];
        if (cc == 'A') nv[c * 5] = 0;
        else if (cc == 'C') nv[c * 5 + 1] = 0;
        else if (cc == 'G') nv[c * 5 + 2] = 0;
        else if (cc == 'T') nv[c * 5 + 3] = 0;
        else if (cc == 'N') nv[c * 5 + 4] = 0;
        else if (cc == '-') nv[c * 5 + 4] = 0;
    }
}

/* R/performPopulationStuctureCorrection.R:49-76: shared counts by direct comparison (integer exact),
 * hdw[j] = 1/(#{i: L - shared[i][j] < thresh} + 1).  states: [L][N]. shared_out may be NULL. */
void orc_hamming_weights(const uint8_t *states, int64_t L, int64_t N, int32_t thresh, double *hdw, int32_t *shared_out,
                         int ncores) {
    int32_t *shared = shared_out ? shared_out : (int32_t *)calloc((size_t)N * N, sizeof(int32_t));
    if (shared_out) memset(shared, 0, (size_t)N * N * sizeof(int32_t));
    /* sequence-major copy so the inner loop streams */
    uint8_t *T = (uint8_t *)malloc((size_t)L * N);
    for (int64_t a = 0; a < L; a++)
        for (int64_t s = 0; s < N; s++) T[s * L + a] = states[a * N + s];
#pragma omp parallel for schedule(dynamic, 4) num_threads(ncores)
    for (int64_t i = 0; i < N; i++)
        for (int64_t j = i; j < N; j++) {
            const uint8_t *x = T + i * L, *y = T + j * L;
            int32_t c = 0;
            for (int64_t a = 0; a < L; a++) c += (x[a] == y[a]);
            shared[i * N + j] = c;
            shared[j * N + i] = c;
        }
    for (int64_t j = 0; j < N; j++) {
        int64_t cnt = 0;
        for (int64_t i = 0; i < N; i++) cnt += ((L - (int64_t)shared[i * N + j]) < thresh);
        hdw[j] = 1.0 / ((double)cnt + 1.0);
    }
    free(T);
    if (!shared_out) free(shared);
}

/*
 * One block of perform_MI_computation_ACGTN (R/computePairwiseMI.R:204-298) with computeMI_Sprase (:390-398):
 * for every state pair (X,Y): pxy = tcrossprod(tXh, CSR(tYh)) + 0.5 (dense rows of sqrt(w)-scaled one-hots
 * times the sparse rows of the other side), then the fused Hadamard pass of src/computeMI.cpp:19 over the
 * LINEAR index, with rft = t(rf rt') * 0.25 read as if it were nf x nt (quirk Q1).
 * MI_out: nf*nt doubles, column-major.  from_idx/to_idx 0-based.
 */
void orc_mi_block(const uint8_t *states, int64_t L, int64_t N, const double *hdw, const double *r, const double *uqe,
                  const int32_t *from_idx, int64_t nf, const int32_t *to_idx, int64_t nt, double *MI_out, int ncores) {
    (void)L;
    double neff = 0;
    {
        long double acc = 0;
        for (int64_t s = 0; s < N; s++) acc += hdw[s];
        neff = (double)acc;
    }
    double *sq = (double *)malloc((size_t)N * sizeof(double));
    for (int64_t s = 0; s < N; s++) sq[s] = sqrt(hdw[s]);
    /* dense from-side matrices tXfh[X] (nf x N, row-major) and marginals pXf = rowSums(tXfh^2) */
    double *tXfh[5], *pXf[5], *pYt[5];
    for (int X = 0; X < 5; X++) {
        tXfh[X] = (double *)calloc((size_t)nf * N, sizeof(double));
        pXf[X] = (double *)calloc((size_t)nf, sizeof(double));
        pYt[X] = (double *)calloc((size_t)nt, sizeof(double));
    }
#pragma omp parallel for num_threads(ncores)
    for (int64_t a = 0; a < nf; a++) {
        const uint8_t *row = states + (int64_t)from_idx[a] * N;
        long double acc[5] = {0, 0, 0, 0, 0};
        for (int64_t s = 0; s < N; s++) {
            int X = row[s];
            if (X < 5) {
                tXfh[X][a * N + s] = sq[s];
                acc[X] += (long double)(sq[s] * sq[s]);
            }
        }
        for (int X = 0; X < 5; X++) pXf[X][a] = (double)acc[X];
    }
    /* to-side in CSR form per state: column indices (values are sq[col]) */
    int64_t *csr_ptr[5];
    int32_t *csr_col[5];
    for (int Y = 0; Y < 5; Y++) {
        csr_ptr[Y] = (int64_t *)calloc((size_t)nt + 1, sizeof(int64_t));
        for (int64_t b = 0; b < nt; b++) {
            const uint8_t *row = states + (int64_t)to_idx[b] * N;
            int64_t c = 0;
            for (int64_t s = 0; s < N; s++) c += (row[s] == Y);
            csr_ptr[Y][b + 1] = csr_ptr[Y][b] + c;
        }
        csr_col[Y] = (int32_t *)malloc((size_t)(csr_ptr[Y][nt] + 1) * sizeof(int32_t));
        for (int64_t b = 0; b < nt; b++) {
            const uint8_t *row = states + (int64_t)to_idx[b] * N;
            int64_t w = csr_ptr[Y][b];
            long double acc = 0;
            for (int64_t s = 0; s < N; s++)
                if (row[s] == Y) {
                    csr_col[Y][w++] = (int32_t)s;
                    acc += (long double)(sq[s] * sq[s]);
                }
            pYt[Y][b] = (double)acc;
        }
    }
    double *pxy = (double *)malloc((size_t)nf * nt * sizeof(double));
    memset(MI_out, 0, (size_t)nf * nt * sizeof(double));
    for (int X = 0; X < 5; X++)
        for (int Y = 0; Y < 5; Y++) {
            /* pxy_t = tcrossprod(tX, CSR(tY)) + 0.5                                           (:391) */
#pragma omp parallel for schedule(static) num_threads(ncores)
            for (int64_t b = 0; b < nt; b++) {
                const int32_t *cols = csr_col[Y] + csr_ptr[Y][b];
                const int64_t nnz = csr_ptr[Y][b + 1] - csr_ptr[Y][b];
                for (int64_t a = 0; a < nf; a++) {
                    const double *xr = tXfh[X] + a * N;
                    double acc = 0;
                    for (int64_t k = 0; k < nnz; k++) acc += xr[cols[k]] * sq[cols[k]];
                    pxy[a + b * nf] = acc + 0.5;
                }
            }
            /* uq_t, pXrX, pYrY, pxpy_tt and .fastHadamard fused                                (:392-396) */
#pragma omp parallel for schedule(static) num_threads(ncores)
            for (int64_t c = 0; c < nf * nt; c++) {
                const int64_t a = c % nf, b = c / nf;
                const double ra = r[from_idx[a]], rb = r[to_idx[b]];
                const double den = neff + ra * rb * 0.5;                                   /* :260 */
                const double RXY = r[from_idx[c / nt]] * r[to_idx[c % nt]] * 0.25;        /* :261 + linear index (Q1) */
                const double uq = uqe[(int64_t)from_idx[a] * 5 + X] * uqe[(int64_t)to_idx[b] * 5 + Y];
                const double pX = pXf[X][a], pY = pYt[Y][b];
                const double pXrX = pX * (0.5 * ra), pYrY = pY * (0.5 * rb), pxpy = pX * pY;
                MI_out[c] += uq * pxy[c] / den * log(pxy[c] / (pxpy + RXY + pXrX + pYrY) * den);
            }
        }
    free(pxy);
    free(sq);
    for (int X = 0; X < 5; X++) {
        free(tXfh[X]);
        free(pXf[X]);
        free(pYt[X]);
        free(csr_ptr[X]);
        free(csr_col[X]);
    }
}

int orc_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
