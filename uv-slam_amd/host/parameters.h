// parameters.h -- mirror of vins_estimator/src/parameters.h:11-66 (the globals Estimator::optimization() reads).
#pragma once
#include <vector>
#include "eigen_lite.h"
const int WINDOW_SIZE = 10;      // parameters.h:12
const int NUM_OF_CAM = 1;
const int NUM_OF_F = 1000;
const int NUM_OF_LF = 1000;
extern double FOCAL_LENGTH, INIT_DEPTH, MIN_PARALLAX, ACC_N, ACC_W, GYR_N, GYR_W, SOLVER_TIME, TD, TR, LINE_FACTOR, VP_FACTOR, ROW, COL;
extern int ESTIMATE_EXTRINSIC, ESTIMATE_TD, NUM_ITERATIONS, LINE_WINDOW;
extern std::vector<Eigen::Matrix3d> RIC;
extern std::vector<Eigen::Vector3d> TIC;
extern Eigen::Vector3d G;
enum SIZE_PARAMETERIZATION { SIZE_POSE = 7, SIZE_SPEEDBIAS = 9, SIZE_FEATURE = 1, SIZE_LINE_FEATURE = 4 };
enum StateOrder { O_P = 0, O_R = 3, O_V = 6, O_BA = 9, O_BG = 12 };
enum NoiseOrder { O_AN = 0, O_GN = 3, O_AW = 6, O_GW = 9 };
void setEurocParameters();       // the values of config/euroc/euroc_config.yaml (the reference reads them through OpenCV FileStorage)
