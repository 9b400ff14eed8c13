// opencv2/imgproc.hpp -- DECLARATION-ONLY stand-in (see core.hpp next to it: not OpenCV, pins nothing, compile check only).
// cv::CLAHE as OpenCV 4.4 declares it (imgproc.hpp): six pure virtuals on top of cv::Algorithm; and the two functions the
// reference's prebuilt binaries import from libopencv_imgproc.so.4.4.
#ifndef MI_TEST_OPENCV_DECL_IMGPROC_HPP_
#define MI_TEST_OPENCV_DECL_IMGPROC_HPP_
#include "core.hpp"
namespace cv {
class CLAHE : public Algorithm {
public:
    virtual void apply(InputArray src, OutputArray dst) = 0;
    virtual void setClipLimit(double clipLimit) = 0;
    virtual double getClipLimit() const = 0;
    virtual void setTilesGridSize(Size tileGridSize) = 0;
    virtual Size getTilesGridSize() const = 0;
    virtual void collectGarbage() = 0;
};
Ptr<CLAHE> createCLAHE(double clipLimit = 40.0, Size tileGridSize = Size(8, 8));
void equalizeHist(InputArray src, OutputArray dst);
enum ColorConversionCodes { COLOR_BGR2YUV = 82, COLOR_YUV2BGR = 84, COLOR_YUV2BGR_NV12 = 91, COLOR_BGR2YUV_I420 = 128 };
void cvtColor(InputArray src, OutputArray dst, int code, int dstCn = 0);
}  // namespace cv
#endif
