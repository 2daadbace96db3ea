"""Host-side mirror of the reference's triangulation entry point.

Reconstruction::triangulateCloud (Header Files/reconstruction.h:154-156,
Source Files/reconstruction.cpp:41-73) and CameraParameters
(Header Files/camera.h:11-42), with NumPy arrays standing in for cv::Matx and
std::vector.
"""
import numpy as np

from . import capi
from .feature_processing import default_context


class CameraParameters:
    """Header Files/camera.h:21-42: K scaled by the down-sampling factor, K(2,2) = 1."""

    def __init__(self, K, distCoeffs=None, downSample=1.0):
        self.updateCameraParameters(K, distCoeffs, downSample)

    def updateCameraParameters(self, K, distCoeffs=None, downSample=1.0):
        K = np.asarray(K, dtype=np.float64).reshape(3, 3) / float(downSample)
        K[2, 2] = 1.0
        self.K = K
        self.K33d = K.copy()
        self.distCoeffs = distCoeffs
        self.pp = (K[0, 2], K[1, 2])
        self.focal = (K[0, 0], K[1, 1])


class Reconstruction:
    """Source Files/reconstruction.cpp:3-73 (bundle adjustment is not part of this path)."""

    def __init__(self, triangulateMethod, baMethod="", baMaxRMSE=0.0, minDistance=1e-4,
                 maxDistance=250.0, maxProjectionError=3.0, useNormalizePts=True, ctx=None):
        self.m_triangulateMethod = str(triangulateMethod)
        self.m_baMethod = baMethod
        self.m_baMaxRMSE = baMaxRMSE
        self.m_minDistance = np.float32(minDistance)
        self.m_maxDistance = np.float32(maxDistance)
        self.m_maxProjectionError = np.float32(maxProjectionError)
        self.m_useNormalizePts = bool(useNormalizePts)
        self._ctx = ctx

    @property
    def ctx(self):
        return self._ctx if self._ctx is not None else default_context()

    def triangulateCloud(self, camera, prevPts, currPts, colorImage, prevPose, currPose, R, t):
        """Returns (points3D [n,3] float64, pointsRGB [n,3] uint8, mask [n] bool).

        As in the reference, empty input returns empty output (reconstruction.cpp:42), R and t
        are the separately supplied pose used for reprojection (they are not derived from
        currPose), and the undistortion with empty coefficients is an identity.
        """
        if self.m_triangulateMethod == "DLT":
            # reconstruction.cpp:59-66: the only call in that branch is commented out
            raise capi.SfmError('triangulateMethod "DLT" is a dead branch in the reference; '
                                'use "ITERATIVE" (cv::triangulatePoints)')
        prevPts = np.asarray(prevPts, dtype=np.float32).reshape(-1, 2)
        currPts = np.asarray(currPts, dtype=np.float32).reshape(-1, 2)
        if prevPts.shape[0] == 0 or currPts.shape[0] == 0:
            return np.zeros((0, 3)), np.zeros((0, 3), np.uint8), np.zeros(0, bool)
        X, rgb, mask, _ = self.ctx.triangulate(
            camera.K33d, prevPose, currPose, R, t, prevPts, currPts, bgr=colorImage,
            min_dist=float(self.m_minDistance), max_dist=float(self.m_maxDistance),
            max_proj_err=float(self.m_maxProjectionError))
        return X, rgb, mask.astype(bool)
