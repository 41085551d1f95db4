/*
 * rfx_host_api.h -- C ABI of librfxhost.so: the HOST side of the exact RANSAC mode.
 *
 * The reference solves every 4-point DLT system with the host's LAPACK (utils/outil.py:68-87: float32 products stored
 * into a float64 8x9 matrix, np.linalg.svd, row 8 of Vh, .float()).  On a full-rank system the device's Householder null
 * vector equals LAPACK's to the last bit after the float32 cast (csrc/dlt.h); on a RANK-DEFICIENT system (three of the
 * four matched points collinear in both images) the null space is two-dimensional and "the reference's value" is whatever
 * the host's dgesdd returns.  Those systems -- flagged on the device, gathered into pinned host memory by
 * rfx_ransac_degenerate_gather (rfx_api.h) -- are re-solved here, by the SAME dgesdd routine of the SAME shared object
 * numpy's _umath_linalg module is linked to, called the way numpy's gufunc calls it (jobz 'A', M = 8, N = 9, column-major
 * copy, work size from a query), on plain std::threads: no Python, no GIL, no pipes.
 *
 * Plain pointers and sizes; every pointer is a HOST pointer.  Returns 0 on success, RFX_HOST_E_* (< 0) otherwise.
 * Nothing here touches a GPU: the library links against libdl / libpthread only and loads on a GPU-less machine.
 */
#ifndef RFX_HOST_API_H
#define RFX_HOST_API_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RFX_HOST_OK 0
#define RFX_HOST_E_ARG (-1)        /* null pointer / negative size */
#define RFX_HOST_E_UNBOUND (-2)    /* rfx_host_lapack_bind has not succeeded */
#define RFX_HOST_E_LAPACK (-3)     /* dgesdd returned info != 0 (the reference would raise LinAlgError) */

/* Resolve dgesdd through the shared object `module_path` (numpy.linalg._umath_linalg.__file__): dlopen'ed with
 * RTLD_NOLOAD first -- the object numpy has already mapped, so that dlsym() walks ITS dependency list and finds the routine
 * its own svd gufunc binds (scipy_dgesdd_64_ of the bundled OpenBLAS, dgesdd_64_, dgesdd_ ...; a numpy built on its f2c
 * lapack_lite defines dgesdd_ itself).  Returns the width in bits of the routine's INTEGER arguments (32 or 64), or
 * RFX_HOST_E_ARG when no candidate symbol resolves.  Idempotent; rfx_host_lapack_symbol() names what was bound. */
int rfx_host_lapack_bind(const char* module_path);
const char* rfx_host_lapack_symbol(void);

/* Size of the worker pool used by rfx_host_dlt_null_vectors (the calling thread works too).  n <= 0: leave unchanged.
 * Returns the pool size in effect.  Default: min(hardware threads, 64). */
int rfx_host_set_threads(int n);

/* utils/outil.py:68-87 for k systems.  xy: k rows of 16 floats -- the 4 source points (u', v') then the 4 target points
 * (u, v) of a sample, i.e. X[:, :, :2] | Y[:, :, :2] of outil.Homography(X, Y).  For every row: rows 2i / 2i+1 of the 8x9
 * system are [0,0,0,-u,-v,-1, v'u, v'v, v'] / [u,v,1,0,0,0, -u'u, -u'v, -u'] with the PRODUCTS ROUNDED TO FLOAT32 and stored
 * as float64 (utils/outil.py:72-83), dgesdd(jobz 'A'), h = Vh[8, :].
 * H_out: k x 9 float32 (the reference's .float()), or NULL;  hv_out: k x 9 float64 (Vh[8] itself), or NULL.
 * dedupe != 0: rows with byte-identical content inside a block of 1024 consecutive rows are solved once (a late
 * multi-homography round with 5 matches left draws the same 120 ordered samples over and over); the result is the same
 * bits either way -- one routine, one input, one output.
 * n_solved_out (optional): number of dgesdd calls made. */
int rfx_host_dlt_null_vectors(const float* xy, int64_t k, float* H_out, double* hv_out, int dedupe, int64_t* n_solved_out);

#ifdef __cplusplus
}
#endif
#endif /* RFX_HOST_API_H */
