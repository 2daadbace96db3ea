"""Host-side mirror of the reference's matcher and optical-flow wrappers.

Same class names, method names, argument meaning and error behaviour as
Header Files/feature_processing.h:33-57 (DescriptorMatcher) and :75-100
(OptFlow) of the reference; the bodies call the HIP library through the C ABI
instead of OpenCV.  NumPy arrays stand in for cv::Mat / std::vector:
descriptors are (rows x length) uint8 or float32 arrays, key points are (n x 2)
float32 arrays of KeyPoint::pt, matches are structured arrays with cv::DMatch's
fields.
"""
import numpy as np

from . import capi

_default_ctx = None


def default_context():
    """One lazily created context per process (the reference is single-threaded)."""
    global _default_ctx
    if _default_ctx is None:
        _default_ctx = capi.Context(0)
    return _default_ctx


class FeatureDetector:
    """Source Files/feature_processing.cpp:3-99, the part on the tracker's path: the Shi-Tomasi
    corners generateFlowFeatures hands to OptFlow::computeFlow.  The key point detectors and
    descriptor extractors of generateFeatures (AKAZE, ORB, SIFT ...) are OpenCV's and are not
    provided."""

    def __init__(self, method="AKAZE", ctx=None):
        self.method = str(method).upper()
        self.ctx = ctx if ctx is not None else default_context()

    def generateFeatures(self, imGray):
        raise capi.SfmError("key point detection / description is not part of sfm_hip "
                            "(Source Files/feature_processing.cpp:78-85 stays with OpenCV)")

    def generateFlowFeatures(self, imGray, corners, maxCorners, qualityLevel, minDistance):
        """feature_processing.cpp:87-99: cv::goodFeaturesToTrack, appended to `corners`
        (good points already there are kept).  Returns the extended float32 [n, 2] array."""
        found = self.ctx.good_features(imGray, int(maxCorners), float(qualityLevel), float(minDistance))
        have = np.asarray(corners, dtype=np.float32).reshape(-1, 2)
        return np.concatenate([have, found], axis=0)


class DescriptorMatcher:
    """Source Files/feature_processing.cpp:101-226."""

    def __init__(self, method, ratioThreshold, isVisDebug=False, visDebugWinSize=None, ctx=None):
        # feature_processing.cpp:104-115: upper-case, three named types, else BRUTEFORCE
        method = str(method).upper()
        if method == "BRUTEFORCE_HAMMING":
            self.norm = capi.NORM_HAMMING
        elif method == "BRUTEFORCE_SL2":
            self.norm = capi.NORM_L2SQR
        elif method == "FLANNBASED":
            raise capi.SfmError("FLANNBASED (approximate kd-tree search) is not part of the "
                                "brute-force GPU path")
        else:
            self.norm = capi.NORM_L2
        self.m_ratioThreshold = np.float32(ratioThreshold)
        self.m_isVisDebug = bool(isVisDebug)
        self._ctx = ctx

    @property
    def ctx(self):
        return self._ctx if self._ctx is not None else default_context()

    def knnMatch(self, queryDesc, trainDesc):
        """matcher->knnMatch(l, r, out, 2): ([nq,2] int32 train indices, [nq,2] float32)."""
        return self.ctx.knn2(self.norm, queryDesc, trainDesc)

    def ratioMaches(self, lDesc, rDesc):
        """feature_processing.cpp:118-130; returns the matches (cleared, then filled)."""
        return self.ctx.ratio_matches(self.norm, lDesc, rDesc, float(self.m_ratioThreshold))

    def findRobustMatches(self, prevKeyPts, currKeyPts, prevDesc, currDesc,
                          useEpipolarFilter):
        """feature_processing.cpp:144-226.

        Returns (prevAligPts, currAligPts, matches, prevPtsToKeyIdx, currPtsToKeyIdx), i.e.
        what the reference leaves in its five output vectors: the cross-checked matches, and
        with useEpipolarFilter only those cv::findFundamentalMat marks as inliers (:187-204).
        """
        prevKeyPts = np.asarray(prevKeyPts, dtype=np.float32).reshape(-1, 2)
        currKeyPts = np.asarray(currKeyPts, dtype=np.float32).reshape(-1, 2)
        m = self.ctx.robust_matches(self.norm, prevDesc, currDesc, float(self.m_ratioThreshold))
        if useEpipolarFilter and len(m):
            _, _, mask = self.ctx.fundamental(prevKeyPts[m["queryIdx"]], currKeyPts[m["trainIdx"]])
            m = m[mask.astype(bool)]
        prev_idx = m["queryIdx"].astype(np.int32)
        curr_idx = m["trainIdx"].astype(np.int32)
        return prevKeyPts[prev_idx], currKeyPts[curr_idx], m, prev_idx, curr_idx


class OptFlowAddSettings:
    """Header Files/feature_processing.h:59-73."""

    def __init__(self):
        self.maxError = 0.0
        self.qualLvl = 0.0
        self.minDist = 0.0
        self.maxCorn = 0
        self.minFeatures = 0


class OptFlow:
    """Source Files/feature_processing.cpp:228-249.

    termcrit is (type, maxCount, epsilon) with cv::TermCriteria's type bits
    (COUNT = 1, EPS = 2), as built at Source Files/app_solver.cpp:108.
    """

    def __init__(self, termcrit, winSize, maxLevel, maxError, maxCorners=0, qualityLevel=0.0,
                 minCornersDistance=0.0, minFeatures=0, ctx=None):
        self.termcrit = (int(termcrit[0]), int(termcrit[1]), float(termcrit[2]))
        self.winSize = (int(winSize), int(winSize))
        self.maxLevel = int(maxLevel)
        self.minEigThreshold = 1e-4  # cv::SparsePyrLKOpticalFlow::create default
        self.additionalSettings = OptFlowAddSettings()
        self.additionalSettings.maxError = float(maxError)
        self.additionalSettings.maxCorn = int(maxCorners)
        self.additionalSettings.qualLvl = float(qualityLevel)
        self.additionalSettings.minDist = float(minCornersDistance)
        self.additionalSettings.minFeatures = int(minFeatures)
        self.statusMask = np.zeros(0, np.uint8)
        self._ctx = ctx

    @property
    def ctx(self):
        return self._ctx if self._ctx is not None else default_context()

    def computeFlow(self, imPrevGray, imCurrGray, prevPts):
        """Returns (currPts, statusMask); status is 0 where err > maxError (:245-248)."""
        pts = np.asarray(prevPts, dtype=np.float32).reshape(-1, 2)
        if pts.shape[0] == 0:
            return np.zeros((0, 2), np.float32), np.zeros(0, np.uint8)
        curr, status, _ = self.ctx.pyrlk(
            imPrevGray, imCurrGray, pts, win=self.winSize, max_level=self.maxLevel,
            crit_type=self.termcrit[0], max_count=self.termcrit[1], epsilon=self.termcrit[2],
            min_eig_threshold=self.minEigThreshold, max_error=self.additionalSettings.maxError)
        return curr, status
