// kmeans driver -- command line of the reference's drivers/kmeans.cpp:198-201:
//   kmeans_driver <points> <centers> <npoints> <ndims> <ncenters>
// One Lloyd iteration, as the reference's main() runs (:221-224): squared distances of every
// point to every centre through flash::kmeans (the dist matrix is a flash_malloc'ed file, column-
// major ncenters x npoints: drivers/kmeans.cpp:37-39, 113-114), nearest centre per point, centres
// replaced by the means of their points, written back to the centres file in place.  The
// distance matrix is the flash kernel's work; the reductions around it are plain host loops here
// (the reference uses cblas_sdot / cblas_isamin / cblas_saxpy on the mapped files).
#include <limits>

#include "driver_util.h"

int main(int argc, char** argv) {
  const drv::Args arg(argc, argv, 5, "<points> <centers> <npoints> <ndims> <ncenters>");
  const FBLAS_UINT npoints = arg.u(3), ndims = arg.u(4), ncenters = arg.u(5);
  FBLAS_INT res;
  {
    drv::Session lib("/tmp/kmeans_driver_temps");
    auto points = lib.map<FPTYPE>(arg.str(1)), centers = lib.map<FPTYPE>(arg.str(2));
    std::vector<FPTYPE> P(npoints * ndims), Cn(ncenters * ndims);
    flash::read_sync(P.data(), points, npoints * ndims);
    flash::read_sync(Cn.data(), centers, ncenters * ndims);
    std::vector<FPTYPE> p_l2sq(npoints), c_l2sq(ncenters), ones(std::max(npoints, ncenters), (FPTYPE) 1);
    auto sq = [ndims](const FPTYPE* v) {
      FPTYPE s = 0;
      for (FBLAS_UINT d = 0; d < ndims; d++) s += v[d] * v[d];
      return s;
    };
    for (FBLAS_UINT p = 0; p < npoints; p++) p_l2sq[p] = sq(&P[p * ndims]);
    for (FBLAS_UINT c = 0; c < ncenters; c++) c_l2sq[c] = sq(&Cn[c * ndims]);

    flash::flash_ptr<FPTYPE> dist = flash::flash_malloc<FPTYPE>(ncenters * npoints * sizeof(FPTYPE), "dist_mat");
    flash::Timer timer;
    res = flash::kmeans('C', 'T', 'N', ncenters, npoints, ndims, (FPTYPE) -2.0, (FPTYPE) 0.0, centers, points, dist,
                        ndims, ndims, ncenters, c_l2sq.data(), p_l2sq.data(), ones.data());
    GLOG_INFO("kmeans() took ", timer.elapsed() / 1000);
    GLOG_INFO("flash::kmeans() returned with ", res);

    std::vector<FPTYPE> D(ncenters * npoints);
    flash::read_sync(D.data(), dist, ncenters * npoints);
    flash::flash_free(dist);
    std::vector<double> sum(ncenters * ndims, 0.0);
    std::vector<FBLAS_UINT> count(ncenters, 0);
    double residual = 0.0;
    for (FBLAS_UINT p = 0; p < npoints; p++) {
      FBLAS_UINT best = 0;
      FPTYPE bd = std::numeric_limits<FPTYPE>::max();
      for (FBLAS_UINT c = 0; c < ncenters; c++)
        if (D[p * ncenters + c] < bd) { bd = D[p * ncenters + c]; best = c; }
      residual += bd > 0 ? bd : 0;
      count[best]++;
      for (FBLAS_UINT d = 0; d < ndims; d++) sum[best * ndims + d] += P[p * ndims + d];
    }
    for (FBLAS_UINT c = 0; c < ncenters; c++)
      for (FBLAS_UINT d = 0; d < ndims; d++)
        Cn[c * ndims + d] = count[c] ? (FPTYPE) (sum[c * ndims + d] / (double) count[c]) : (FPTYPE) 0;  // empty cluster: the reference's memset(0)
    flash::write_sync(centers, Cn.data(), ncenters * ndims);
    GLOG_INFO("residual before the update : ", residual);
  }
  return res == 0 ? 0 : 1;
}
