// Declarations-only mirror of the part of OpenCV 4's features2d API the adapter touches
// (see core.hpp in this directory).
#ifndef SFM_TEST_OPENCV_DECL_FEATURES2D_HPP
#define SFM_TEST_OPENCV_DECL_FEATURES2D_HPP
#include <opencv2/core.hpp>

namespace cv {

class Feature2D : public virtual Algorithm {
public:
    virtual ~Feature2D();
    virtual void detect(InputArray image, std::vector<KeyPoint>& keypoints, InputArray mask = noArray());
    virtual void compute(InputArray image, std::vector<KeyPoint>& keypoints, OutputArray descriptors);
    virtual void detectAndCompute(InputArray image, InputArray mask, std::vector<KeyPoint>& keypoints,
                                  OutputArray descriptors, bool useProvidedKeypoints = false);
    virtual int descriptorSize() const;
    virtual int descriptorType() const;
    virtual int defaultNorm() const;
};
typedef Feature2D FeatureDetector;
typedef Feature2D DescriptorExtractor;

enum NormTypes { NORM_INF = 1, NORM_L1 = 2, NORM_L2 = 4, NORM_L2SQR = 5, NORM_HAMMING = 6, NORM_HAMMING2 = 7 };

class DescriptorMatcher : public Algorithm {
public:
    enum MatcherType { FLANNBASED = 1, BRUTEFORCE = 2, BRUTEFORCE_L1 = 3, BRUTEFORCE_HAMMING = 4,
                       BRUTEFORCE_HAMMINGLUT = 5, BRUTEFORCE_SL2 = 6 };
    virtual ~DescriptorMatcher();
    virtual void add(InputArrayOfArrays descriptors);
    const std::vector<Mat>& getTrainDescriptors() const;
    virtual void clear() override;
    virtual bool empty() const override;
    virtual bool isMaskSupported() const = 0;
    virtual void train();
    void knnMatch(InputArray queryDescriptors, InputArray trainDescriptors, std::vector<std::vector<DMatch>>& matches,
                  int k, InputArray mask = noArray(), bool compactResult = false) const;
    void knnMatch(InputArray queryDescriptors, std::vector<std::vector<DMatch>>& matches, int k,
                  InputArrayOfArrays masks = noArray(), bool compactResult = false);
    virtual Ptr<DescriptorMatcher> clone(bool emptyTrainData = false) const = 0;
    static Ptr<DescriptorMatcher> create(const String& descriptorMatcherType);

protected:
    virtual void knnMatchImpl(InputArray queryDescriptors, std::vector<std::vector<DMatch>>& matches, int k,
                              InputArrayOfArrays masks = noArray(), bool compactResult = false) = 0;
    virtual void radiusMatchImpl(InputArray queryDescriptors, std::vector<std::vector<DMatch>>& matches,
                                 float maxDistance, InputArrayOfArrays masks = noArray(),
                                 bool compactResult = false) = 0;
    std::vector<Mat> trainDescCollection;
};

}  // namespace cv
#endif
