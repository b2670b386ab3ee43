// TwoViewReconstruct -- drop-in driver for the reference's TwoViewReconstruct.cpp main() (lines 50-97; the file is
// commented out in the reference but is the L2 / SIFT twin BASELINE.json's configs[0] names): match two images,
// essential matrix, homogeneous triangulation, structure.yml with float points.
#include "sfm_pipeline.hpp"

int main(int argc, char** argv) { return sfm::driver_main(argc, argv, false); }
