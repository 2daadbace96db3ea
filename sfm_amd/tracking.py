"""Host-side mirror of the reference's relative-pose step.

RecoveryPose (Header Files/tracking.h:17-32, Source Files/tracking.cpp:3-18) and
Tracking::findCameraPose (Header Files/tracking.h:82, Source Files/tracking.cpp:91-113), with
NumPy arrays standing in for cv::Mat / cv::Matx and std::vector.  The PnP tracking of
findRecoveredCameraPose and the drawing helpers are not part of the GPU path.
"""
import numpy as np

from . import capi
from .feature_processing import default_context


class RecoveryPose:
    """Pose estimation settings and result.  `method` is "RANSAC" or "LMEDS" (anything else
    selects RANSAC, as the reference's string mapping does)."""

    def __init__(self, method="RANSAC", prob=0.99, threshold=0.5, minInliers=10, poseEstMethod="SOLVEPNP_P3P",
                 useExtrinsicGuess=False, numIter=100):
        self.recPoseMethod = capi.POSE_LMEDS if str(method).upper() == "LMEDS" else capi.POSE_RANSAC
        self.poseEstMethod = poseEstMethod
        self.prob = float(prob)
        self.threshold = float(threshold)
        self.minInliers = int(minInliers)
        self.numIter = int(numIter)
        self.useExtrinsicGuess = bool(useExtrinsicGuess)
        self.R = np.eye(3)
        self.t = np.zeros(3)
        self.mask = np.zeros(0, np.uint8)


class Tracking:
    @staticmethod
    def findCameraPose(recPose, prevPts, currPts, cameraK, minInliers, ctx=None):
        """Source Files/tracking.cpp:91-113.  Returns (ok, numInliers); R, t and mask are left in
        recPose as the reference leaves them."""
        ctx = ctx if ctx is not None else default_context()
        prevPts = np.asarray(prevPts, dtype=np.float32).reshape(-1, 2)
        currPts = np.asarray(currPts, dtype=np.float32).reshape(-1, 2)
        if len(prevPts) <= 5 or len(currPts) <= 5:
            return False, 0
        E, mask = ctx.find_essential(prevPts, currPts, cameraK, recPose.recPoseMethod, recPose.prob,
                                     recPose.threshold)
        recPose.mask = mask
        if len(E) != 1:  # the reference's "E.cols == 3 && E.rows == 3"
            return False, 0
        K = np.asarray(cameraK, dtype=np.float64).reshape(3, 3)
        Kinv = np.linalg.inv(K)
        p0 = np.array([prevPts[0, 0], prevPts[0, 1], 1.0])
        p1 = np.array([currPts[0, 0], currPts[0, 1], 1.0])
        if abs(p1 @ Kinv.T @ E[0] @ Kinv @ p0) > 1e-3:
            return False, 0
        numInliers, recPose.R, recPose.t, recPose.mask = ctx.recover_pose(E[0], prevPts, currPts, K, mask)
        return numInliers > minInliers, numInliers
