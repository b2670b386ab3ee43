// NViewReconstruct -- drop-in driver for the reference's NViewReconstuct.cpp main() (lines 1334-1524): the same stage
// order, stdout lines and output files (structure.yml, structure_ba.yml, structure_ba.ply; pre-BA poses in
// structure_ba.yml like the reference unless --write-back-poses), with matching / triangulation / bundle adjustment /
// normals on libsfmhip.so.  Starts from a features file instead of a .jpg directory (see sfm_pipeline.hpp).
#include "sfm_pipeline.hpp"

int main(int argc, char** argv) { return sfm::driver_main(argc, argv, true); }
