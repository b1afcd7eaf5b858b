/* sparse_ldlt.c -- sparse LDL' without pivoting on the CPU: ordering, symbolic and numeric phases, solves.
 * TEST INFRASTRUCTURE ONLY (see sqp_oracle.h): the checker of the product's multifrontal path.
 *
 * Stands where the reference has Ipopt's linear solver (MUMPS / MA57, not vendored:
 * /root/reference/examples/acopf/opf.jl:59-64) behind JuMP.optimize!
 * (/root/reference/src/algorithms/subproblem_JuMP.jl:178).  Written independently of the product's symbolic.hip /
 * mfront.hip so that the two can check each other:
 *   ordering  plain minimum degree on the explicit elimination graph (exact degrees, sorted adjacency arrays, a binary
 *             heap with lazy deletion; ties by lowest index), with the same precedence rule as the product -- a row of
 *             the quasi-definite matrix is eliminated only behind every variable it couples to (a row pivoted before
 *             its variables has a pivot of the size of the 1e-8 regularisation);
 *   factor    the up-looking algorithm of T. A. Davis, "Algorithm 849: a concise sparse Cholesky factorization
 *             package", ACM TOMS 31 (2005): elimination tree and column counts by tree ascent, then row k of L by a
 *             sparse triangular solve over the reach of column k in the tree.  Restated from the paper.
 */
#include "sqp_oracle.h"
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

struct ora_sldl {
    int64_t n, nt, nnzU, nnzL;
    int64_t *perm, *pinv;           /* position -> index, index -> position */
    int64_t *Up, *Ui, *slot;        /* permuted upper triangle (rows <= column) in CSC; triplet -> entry */
    double *Ux;
    int64_t *Parent, *Lp, *Li, *Lnz, *Flag, *Pattern;
    double *Lx, *D, *Y;
};

/* ---- ordering ------------------------------------------------------------------------------------------------- */
typedef struct { int64_t deg, v; } hent;
typedef struct { hent *a; int64_t len, cap; } heap;
static int hless(hent x, hent y) { return x.deg < y.deg || (x.deg == y.deg && x.v < y.v); }
static void hpush(heap *h, hent e)
{
    if (h->len == h->cap) { h->cap = h->cap ? 2 * h->cap : 1024; h->a = (hent *)realloc(h->a, sizeof(hent) * (size_t)h->cap); }
    int64_t i = h->len++;
    while (i > 0 && hless(e, h->a[(i - 1) / 2])) { h->a[i] = h->a[(i - 1) / 2]; i = (i - 1) / 2; }
    h->a[i] = e;
}
static hent hpop(heap *h)
{
    hent top = h->a[0], e = h->a[--h->len];
    int64_t i = 0;
    for (;;) {
        int64_t c = 2 * i + 1;
        if (c >= h->len) break;
        if (c + 1 < h->len && hless(h->a[c + 1], h->a[c])) ++c;
        if (!hless(h->a[c], e)) break;
        h->a[i] = h->a[c]; i = c;
    }
    if (h->len) h->a[i] = e;
    return top;
}

/* adjacency in CSR (symmetric, no self loops, sorted); before lists in CSR (may be NULL) */
static void min_degree(int64_t n, const int64_t *ap, const int64_t *ai, const int64_t *bp, const int64_t *bi,
                       int64_t *perm)
{
    int64_t **adj = (int64_t **)malloc(sizeof(int64_t *) * (size_t)(n + 1));
    int64_t *len = (int64_t *)malloc(sizeof(int64_t) * (size_t)(n + 1));
    int64_t *need = (int64_t *)calloc((size_t)(n + 1), sizeof(int64_t));
    char *gone = (char *)calloc((size_t)(n + 1), 1);
    int64_t *afp = (int64_t *)calloc((size_t)(n + 2), sizeof(int64_t)), *afi = NULL;
    for (int64_t v = 0; v < n; ++v) {
        len[v] = ap[v + 1] - ap[v];
        adj[v] = (int64_t *)malloc(sizeof(int64_t) * (size_t)(len[v] + 1));
        memcpy(adj[v], ai + ap[v], sizeof(int64_t) * (size_t)len[v]);
    }
    if (bp) {       /* after[v] = the rows waiting for v */
        const char *wk = getenv("ORA_ROWS_AFTER");
        const int weak = wk && atoi(wk) == 2;     /* experiment: a row is eligible behind ANY ONE of its variables */
        for (int64_t u = 0; u < n; ++u) { need[u] = bp[u + 1] - bp[u]; if (weak) { const char *sl = getenv("ORA_ROWS_SLACK"); if (sl) { if (need[u] > 0) { need[u] -= atoi(sl); if (need[u] < 1) need[u] = 1; } } else if (need[u] > 1) need[u] = 1; } for (int64_t k = bp[u]; k < bp[u + 1]; ++k) afp[bi[k] + 2]++; }
        for (int64_t v = 0; v < n; ++v) afp[v + 2] += afp[v + 1];
        afi = (int64_t *)malloc(sizeof(int64_t) * (size_t)(bp[n] + 1));
        for (int64_t u = 0; u < n; ++u) for (int64_t k = bp[u]; k < bp[u + 1]; ++k) afi[afp[bi[k] + 1]++] = u;
    }
    heap H = {0, 0, 0};
    for (int64_t v = 0; v < n; ++v) if (need[v] == 0) { hent e = {len[v], v}; hpush(&H, e); }
    int64_t *tmp = (int64_t *)malloc(sizeof(int64_t) * (size_t)(n + 1));
    for (int64_t k = 0; k < n; ++k) {
        int64_t v = -1;
        while (H.len) { hent e = hpop(&H); if (!gone[e.v] && e.deg == len[e.v] && need[e.v] == 0) { v = e.v; break; } }
        if (v < 0) { fprintf(stderr, "oracle: min_degree: no eligible vertex\n"); abort(); }
        perm[k] = v; gone[v] = 1;
        const int64_t *Nv = adj[v]; const int64_t nv = len[v];
        for (int64_t q = 0; q < nv; ++q) {
            const int64_t u = Nv[q];
            /* adj[u] <- (adj[u] \ {v}) U (Nv \ {u}), both sorted */
            const int64_t *A = adj[u]; const int64_t la = len[u];
            int64_t i = 0, j = 0, o = 0;
            while (i < la || j < nv) {
                int64_t x;
                if (j >= nv || (i < la && A[i] < Nv[j])) x = A[i++];
                else if (i >= la || Nv[j] < A[i]) x = Nv[j++];
                else { x = A[i]; ++i; ++j; }
                if (x != v && x != u) tmp[o++] = x;
            }
            free(adj[u]);
            adj[u] = (int64_t *)malloc(sizeof(int64_t) * (size_t)(o + 1));
            memcpy(adj[u], tmp, sizeof(int64_t) * (size_t)o);
            len[u] = o;
            if (need[u] == 0) { hent e = {o, u}; hpush(&H, e); }
        }
        if (bp)
            for (int64_t t = afp[v]; t < afp[v + 1]; ++t) { const int64_t u = afi[t]; if (gone[u] || need[u] <= 0) continue; if (--need[u] == 0) { hent e = {len[u], u}; hpush(&H, e); } }
        free(adj[v]); adj[v] = NULL;
    }
    for (int64_t v = 0; v < n; ++v) free(adj[v]);
    free(adj); free(len); free(need); free(gone); free(afp); free(afi); free(H.a); free(tmp);
}

/* ---- analysis --------------------------------------------------------------------------------------------------- */
typedef struct { int64_t c, r, t; } trip;
static int trip_cmp(const void *a, const void *b)
{
    const trip *x = (const trip *)a, *y = (const trip *)b;
    if (x->c != y->c) return x->c < y->c ? -1 : 1;
    if (x->r != y->r) return x->r < y->r ? -1 : 1;
    return x->t < y->t ? -1 : (x->t > y->t);
}

ora_sldl *ora_sldl_analyse(int64_t n, int64_t nt, const int64_t *ti, const int64_t *tj, const int64_t *bp,
                           const int64_t *bi, int natural)
{
    ora_sldl *S = (ora_sldl *)calloc(1, sizeof(ora_sldl));
    S->n = n; S->nt = nt;
    S->perm = (int64_t *)malloc(sizeof(int64_t) * (size_t)(n + 1));
    S->pinv = (int64_t *)malloc(sizeof(int64_t) * (size_t)(n + 1));
    {   /* symmetric adjacency of the triplets */
        trip *e = (trip *)malloc(sizeof(trip) * (size_t)(2 * nt + 1));
        int64_t ne = 0;
        for (int64_t t = 0; t < nt; ++t)
            if (ti[t] != tj[t]) { e[ne].c = ti[t]; e[ne].r = tj[t]; e[ne++].t = 0; e[ne].c = tj[t]; e[ne].r = ti[t]; e[ne++].t = 0; }
        qsort(e, (size_t)ne, sizeof(trip), trip_cmp);
        int64_t *ap = (int64_t *)calloc((size_t)(n + 2), sizeof(int64_t)), *ai = (int64_t *)malloc(sizeof(int64_t) * (size_t)(ne + 1));
        int64_t o = 0;
        for (int64_t k = 0; k < ne; ++k) {
            if (k && e[k].c == e[k - 1].c && e[k].r == e[k - 1].r) continue;
            ai[o++] = e[k].r; ap[e[k].c + 1]++;
        }
        for (int64_t v = 0; v < n; ++v) ap[v + 1] += ap[v];
        if (natural) for (int64_t v = 0; v < n; ++v) S->perm[v] = v;
        else min_degree(n, ap, ai, bp, bi, S->perm);
        free(e); free(ap); free(ai);
    }
    for (int64_t k = 0; k < n; ++k) S->pinv[S->perm[k]] = k;
    {   /* permuted upper triangle, CSC, duplicates merged; slot of every triplet */
        trip *e = (trip *)malloc(sizeof(trip) * (size_t)(nt + 1));
        for (int64_t t = 0; t < nt; ++t) {
            int64_t a = S->pinv[ti[t]], b = S->pinv[tj[t]];
            e[t].c = a > b ? a : b; e[t].r = a > b ? b : a; e[t].t = t;
        }
        qsort(e, (size_t)nt, sizeof(trip), trip_cmp);
        S->Up = (int64_t *)calloc((size_t)(n + 2), sizeof(int64_t));
        S->Ui = (int64_t *)malloc(sizeof(int64_t) * (size_t)(nt + 1));
        S->slot = (int64_t *)malloc(sizeof(int64_t) * (size_t)(nt + 1));
        int64_t o = 0;
        for (int64_t k = 0; k < nt; ++k) {
            if (!(k && e[k].c == e[k - 1].c && e[k].r == e[k - 1].r)) { S->Ui[o++] = e[k].r; S->Up[e[k].c + 1]++; }
            S->slot[e[k].t] = o - 1;
        }
        for (int64_t v = 0; v < n; ++v) S->Up[v + 1] += S->Up[v];
        S->nnzU = o;
        S->Ux = (double *)calloc((size_t)(o + 1), sizeof(double));
        free(e);
    }
    /* elimination tree and column counts (Davis, ldl_symbolic) */
    S->Parent = (int64_t *)malloc(sizeof(int64_t) * (size_t)(n + 1));
    S->Lnz = (int64_t *)calloc((size_t)(n + 1), sizeof(int64_t));
    S->Flag = (int64_t *)malloc(sizeof(int64_t) * (size_t)(n + 1));
    S->Lp = (int64_t *)calloc((size_t)(n + 2), sizeof(int64_t));
    for (int64_t k = 0; k < n; ++k) {
        S->Parent[k] = -1; S->Flag[k] = k;
        for (int64_t p = S->Up[k]; p < S->Up[k + 1]; ++p)
            for (int64_t i = S->Ui[p]; S->Flag[i] != k; i = S->Parent[i]) {
                if (S->Parent[i] == -1) S->Parent[i] = k;
                S->Lnz[i]++; S->Flag[i] = k;
            }
    }
    for (int64_t k = 0; k < n; ++k) S->Lp[k + 1] = S->Lp[k] + S->Lnz[k];
    S->nnzL = S->Lp[n];
    S->Li = (int64_t *)malloc(sizeof(int64_t) * (size_t)(S->nnzL + 1));
    S->Lx = (double *)malloc(sizeof(double) * (size_t)(S->nnzL + 1));
    S->D = (double *)malloc(sizeof(double) * (size_t)(n + 1));
    S->Y = (double *)calloc((size_t)(n + 1), sizeof(double));
    S->Pattern = (int64_t *)malloc(sizeof(int64_t) * (size_t)(n + 1));
    return S;
}

void ora_sldl_free(ora_sldl *S)
{
    if (!S) return;
    void *p[] = { S->perm, S->pinv, S->Up, S->Ui, S->slot, S->Ux, S->Parent, S->Lp, S->Li, S->Lnz, S->Flag, S->Pattern,
                  S->Lx, S->D, S->Y };
    for (size_t i = 0; i < sizeof(p) / sizeof(p[0]); ++i) free(p[i]);
    free(S);
}

int64_t ora_sldl_nnz_l(const ora_sldl *S) { return S->nnzL; }
const int64_t *ora_sldl_perm(const ora_sldl *S) { return S->perm; }
const double *ora_sldl_pivots(const ora_sldl *S) { return S->D; }

/* numeric factorisation from the triplet values (duplicates summed in triplet order); returns the number of
 * positive pivots, *nbad = pivots that are zero or not finite */
int64_t ora_sldl_numeric(ora_sldl *S, const double *tv, int64_t *nbad)
{
    const int64_t n = S->n;
    memset(S->Ux, 0, sizeof(double) * (size_t)S->nnzU);
    for (int64_t t = 0; t < S->nt; ++t) S->Ux[S->slot[t]] += tv[t];
    double *Y = S->Y;
    for (int64_t k = 0; k < n; ++k) {
        /* pattern of row k of L = reach of column k's entries in the tree, in topological order */
        int64_t top = n;
        Y[k] = 0.0; S->Flag[k] = k; S->Lnz[k] = 0;
        for (int64_t p = S->Up[k]; p < S->Up[k + 1]; ++p) {
            int64_t i = S->Ui[p], len = 0;
            Y[i] += S->Ux[p];
            for (; S->Flag[i] != k; i = S->Parent[i]) { S->Pattern[len++] = i; S->Flag[i] = k; }
            while (len > 0) S->Pattern[--top] = S->Pattern[--len];
        }
        double dk = Y[k];
        Y[k] = 0.0;
        for (; top < n; ++top) {
            const int64_t i = S->Pattern[top];
            const double yi = Y[i];
            Y[i] = 0.0;
            const int64_t p2 = S->Lp[i] + S->Lnz[i];
            for (int64_t p = S->Lp[i]; p < p2; ++p) Y[S->Li[p]] -= S->Lx[p] * yi;
            const double lki = yi / S->D[i];
            dk -= lki * yi;
            S->Li[p2] = k; S->Lx[p2] = lki; S->Lnz[i]++;
        }
        S->D[k] = dk;
    }
    int64_t np = 0, bad = 0;
    for (int64_t k = 0; k < n; ++k) {
        const double d = S->D[k];
        if (!isfinite(d) || d == 0.0) ++bad; else if (d > 0) ++np;
    }
    if (nbad) *nbad = bad;
    return np;
}

/* x <- K^-1 x, x in the original index order */
void ora_sldl_solve(const ora_sldl *S, double *x)
{
    const int64_t n = S->n;
    double *y = (double *)malloc(sizeof(double) * (size_t)(n + 1));
    for (int64_t k = 0; k < n; ++k) y[k] = x[S->perm[k]];
    for (int64_t j = 0; j < n; ++j) {
        const double yj = y[j];
        for (int64_t p = S->Lp[j]; p < S->Lp[j] + S->Lnz[j]; ++p) y[S->Li[p]] -= S->Lx[p] * yj;
    }
    for (int64_t j = 0; j < n; ++j) y[j] /= S->D[j];
    for (int64_t j = n - 1; j >= 0; --j) {
        double a = y[j];
        for (int64_t p = S->Lp[j]; p < S->Lp[j] + S->Lnz[j]; ++p) a -= S->Lx[p] * y[S->Li[p]];
        y[j] = a;
    }
    for (int64_t k = 0; k < n; ++k) x[S->perm[k]] = y[k];
    free(y);
}
