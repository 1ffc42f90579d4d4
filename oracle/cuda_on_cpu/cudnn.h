/* cuda_on_cpu/cudnn.h -- TEST INFRASTRUCTURE ONLY: the two enums common/helper.h names */
#ifndef CUDA_ON_CPU_CUDNN_H
#define CUDA_ON_CPU_CUDNN_H
typedef enum { CUDNN_STATUS_SUCCESS = 0, CUDNN_STATUS_BAD_PARAM = 3 } cudnnStatus_t;
typedef enum { CUDNN_DATA_FLOAT = 0, CUDNN_DATA_DOUBLE = 1, CUDNN_DATA_HALF = 2 } cudnnDataType_t;
#endif
