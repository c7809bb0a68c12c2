// Package embeddings: drop-in replacement of imageclust/internal/embeddings (embeddings.go:17-163) backed by
// libimageclust_hip.so.  gocv.Net / gocv.Mat become the opaque Net / Mat below; workflow.go:50-55 stays
// source-compatible (it only stores the value).  NOT COMPILED in the authoring container (no Go toolchain).
package embeddings

/*
#cgo CFLAGS: -I${SRCDIR}/../../../include
#cgo LDFLAGS: -L${SRCDIR}/../../../imageclust_amd -limageclust_hip
#include <stdlib.h>
#include "imageclust.h"
*/
import "C"

import (
	"fmt"
	"sync"
	"unsafe"
)

// Net stands in for gocv.Net: a GPU context holding ResNet50-v1 weights in HBM.
type Net struct{ ctx *C.icl_ctx }

func (n Net) Empty() bool { return n.ctx == nil }
func (n *Net) Close() error {
	if n.ctx != nil {
		C.icl_destroy(n.ctx)
		n.ctx = nil
	}
	return nil
}

// AppContext mirrors embeddings.go:17-25.
type AppContext struct {
	ImageDir      string
	CacheDir      string
	LabelSet      map[string]int
	Mutex         sync.Mutex
	LabelsMapping map[string][]string
	Net           Net
	NetMutex      sync.Mutex
}

// LoadPretrainedModelONNX: embeddings.go:28-43.
func LoadPretrainedModelONNX(modelPath string) (Net, error) {
	var ctx *C.icl_ctx
	if rc := C.icl_create(0, &ctx); rc != C.ICL_OK {
		return Net{}, fmt.Errorf("failed to load ResNet50 ONNX model from: %s (%s)", modelPath, C.GoString(C.icl_last_error(nil)))
	}
	p := C.CString(modelPath)
	defer C.free(unsafe.Pointer(p))
	if rc := C.icl_model_load_onnx(ctx, p); rc != C.ICL_OK {
		msg := C.GoString(C.icl_last_error(ctx))
		C.icl_destroy(ctx)
		return Net{}, fmt.Errorf("failed to load ResNet50 ONNX model from: %s (%s)", modelPath, msg)
	}
	return Net{ctx: ctx}, nil
}

// GetImageEmbedding: embeddings.go:119-163.  The returned slice is Go-owned (the reference's slice aliases a freed
// cv::Mat: embeddings.go:145-152); the engine is thread-safe, NetMutex is kept only for field compatibility.
func GetImageEmbedding(appCtx *AppContext, imagePath string) ([]float32, error) {
	if appCtx.Net.Empty() {
		return nil, fmt.Errorf("failed to generate embedding for image: %s", imagePath)
	}
	out := make([]float32, C.ICL_HEAD_DENSE0) // "resnetv17_dense0_fwd" (embeddings.go:140)
	p := C.CString(imagePath)
	defer C.free(unsafe.Pointer(p))
	if rc := C.icl_embed_file(appCtx.Net.ctx, p, C.ICL_HEAD_DENSE0, (*C.float)(unsafe.Pointer(&out[0]))); rc != C.ICL_OK {
		return nil, fmt.Errorf("%s", C.GoString(C.icl_last_error(appCtx.Net.ctx)))
	}
	return out, nil
}

// GenerateEmbedding is the name BASELINE.json's north_star uses for GetImageEmbedding.
func GenerateEmbedding(appCtx *AppContext, imagePath string) ([]float32, error) {
	return GetImageEmbedding(appCtx, imagePath)
}

// GetImageEmbeddingsBatch is the batched fast path workflow.createEmbeddings should call instead of one goroutine per
// image: images are n*224*224*3 u8 RGB (already resized), the result is n x 2048 (pooled) fp32.
func GetImageEmbeddingsBatch(appCtx *AppContext, rgb []byte, n int) ([]float32, error) {
	out := make([]float32, n*int(C.ICL_HEAD_POOLED))
	if n == 0 {
		return out, nil
	}
	if rc := C.icl_embed_u8(appCtx.Net.ctx, (*C.uint8_t)(unsafe.Pointer(&rgb[0])), C.int64_t(n), C.ICL_HEAD_POOLED, C.ICL_PREC_BF16,
		(*C.float)(unsafe.Pointer(&out[0]))); rc != C.ICL_OK {
		return nil, fmt.Errorf("%s", C.GoString(C.icl_last_error(appCtx.Net.ctx)))
	}
	return out, nil
}
