// cloud_camera.h -- host mirror of the reference's CloudCamera
// (include/agile_grasp2/cloud_camera.h:69-139, src/agile_grasp2/cloud_camera.cpp).
// Same constructors, methods and accessors; value types from types.h.  Container + preprocessing
// only: the heavy work happens behind the C-ABI when the cloud is handed to HandSearch /
// GraspDetector.
#ifndef AGILE_GRASP2_CLOUD_CAMERA_H
#define AGILE_GRASP2_CLOUD_CAMERA_H

#include <string>
#include <vector>

#include "agile_grasp2/messages.h"
#include "agile_grasp2/types.h"

typedef ag2::PointCloudRGB PointCloudRGB;
typedef ag2::PointCloudNormal PointCloudNormal;

class CloudCamera {
 public:
  CloudCamera();
  // cloud_camera.cpp:4-32: cloud with normals; size_left_cloud == size => one camera (source zeros)
  CloudCamera(const PointCloudNormal::Ptr& cloud, int size_left_cloud);
  // cloud_camera.cpp:35-51
  CloudCamera(const PointCloudRGB::Ptr& cloud, int size_left_cloud);
  // cloud_camera.cpp:54-61: PCD file (ASCII or binary, fields x y z [rgb|rgba]); source = ones
  explicit CloudCamera(const std::string& filename);
  // cloud_camera.cpp:64-86: left + right PCD files
  CloudCamera(const std::string& filename_left, const std::string& filename_right);

  // cloud_camera.cpp:89-121.  [minX maxX minY maxY minZ maxZ], strict inequalities.  (The
  // reference's linear indexing of the 2-camera source matrix at :107 is a bug; fixed here.)
  void filterWorkspace(const std::vector<double>& workspace);
  // cloud_camera.cpp:124-168: voxel value = floor((p - min) / cell) * cell + min in float, output
  // sorted lexicographically by (ix, iy, iz); camera source indexed as the reference does it
  // (:137-152: k-th voxel in set order <- k-th first-hit point in scan order).
  void voxelizeCloud(double cell_size);
  // cloud_camera.cpp:171-178 (pcl::RandomSample): num_samples indices without replacement,
  // ascending.  The reference seeds from wall time; here the seed is explicit and the draw is the
  // one ag2_subsample_uniformly makes (include/ag2_c.h).
  void subsampleUniformly(int num_samples, uint64_t seed = 0);
  // Host-side twins of what GraspDetector::preprocessPointCloud runs on the GPU; results identical.
  // Replace the processed cloud with what the GPU front end produced (mirror-only, used by
  // GraspDetector::preprocessPointCloud).
  void adoptProcessed(const PointCloudRGB::Ptr& cloud, const ag2::MatrixXi& camera_source,
                      const ag2::Matrix3Xd& normals);
  // cloud_camera.cpp:181-206
  void subsampleSamples(const agile_grasp2::SamplesMsg& msg, int num_samples, uint64_t seed = 0);

  const ag2::MatrixXi& getCameraSource() const { return camera_source_; }
  const PointCloudRGB::Ptr& getCloudProcessed() const { return cloud_processed_; }
  const PointCloudRGB::Ptr& getCloudOriginal() const { return cloud_original_; }
  const std::vector<int>& getSampleIndices() const { return sample_indices_; }
  const ag2::Matrix3Xd& getNormals() const { return normals_; }
  const ag2::Matrix3Xd& getSamples() const { return samples_; }
  void setSampleIndices(const std::vector<int>& idx) { sample_indices_ = idx; }
  void setSamples(const agile_grasp2::SamplesMsg& msg);
  void setSamples(const ag2::Matrix3Xd& samples) { samples_ = samples; }

  static PointCloudRGB::Ptr loadPointCloudFromFile(const std::string& filename);  // :231-240

 private:
  PointCloudRGB::Ptr cloud_processed_, cloud_original_;
  ag2::MatrixXi camera_source_;       // (i, j) = 1 if point j is seen by camera i
  ag2::Matrix3Xd normals_;            // optional
  std::vector<int> sample_indices_;
  ag2::Matrix3Xd samples_;
};

#endif  // AGILE_GRASP2_CLOUD_CAMERA_H
