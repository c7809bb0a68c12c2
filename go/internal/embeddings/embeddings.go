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
	"runtime"
	"sync"
	"unsafe"

	"imageclust/internal/iclengine"
)

// Net stands in for gocv.Net: the process-wide GPU context (shared with internal/clustering through iclengine) holding
// ResNet50-v1 weights in HBM.  Close releases nothing: the context lives as long as the process, like the reference's model
// which workflow.go:49-50 reloads per request only because nothing caches it.
type Net struct{ ctx *C.icl_ctx }

func (n Net) Empty() bool { return n.ctx == nil }
func (n *Net) Close() error {
	n.ctx = nil
	return nil
}

// Mat stands in for gocv.Mat (embeddings.go:46): the 1x3x224x224 fp32 NCHW blob PreprocessImage returns.
type Mat struct{ Data []float32 }

func (m Mat) Size() []int  { return []int{1, 3, 224, 224} }
func (m Mat) Empty() bool  { return len(m.Data) == 0 }
func (m *Mat) Close() error { m.Data = nil; return nil }

// PreprocessImage: embeddings.go:46-116 -> icl_preprocess_file (IMRead incl. EXIF orientation, cv::resize 224x224
// INTER_LINEAR, BGR->RGB, x/255, NCHW).  The blob is Go-owned.
func PreprocessImage(imagePath string) (Mat, error) {
	out := make([]float32, 3*224*224)
	p := C.CString(imagePath)
	defer C.free(unsafe.Pointer(p))
	runtime.LockOSThread() // icl_last_error(NULL) is thread-local: keep the call and the read on one OS thread
	defer runtime.UnlockOSThread()
	if rc := C.icl_preprocess_file(p, (*C.float)(unsafe.Pointer(&out[0]))); rc != C.ICL_OK {
		return Mat{}, fmt.Errorf("%s", C.GoString(C.icl_last_error(nil)))
	}
	return Mat{Data: out}, nil
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
	raw, err := iclengine.Ctx() // one context per process, shared with internal/clustering
	if err != nil {
		return Net{}, fmt.Errorf("failed to load ResNet50 ONNX model from: %s (%v)", modelPath, err)
	}
	ctx := (*C.icl_ctx)(raw)
	if err := iclengine.LoadModelOnce(modelPath); err != nil { // a repeated load of the same file is a no-op
		return Net{}, fmt.Errorf("failed to load ResNet50 ONNX model from: %s (%v)", modelPath, err)
	}
	return Net{ctx: ctx}, nil
}

// GetImageEmbedding: embeddings.go:119-163.  The returned slice is Go-owned (the reference's slice aliases a freed
// cv::Mat: embeddings.go:145-152).  workflow.go:156-175 calls this from one goroutine per image: icl_embed_file decodes and
// resizes on the calling thread and COALESCES the concurrent callers into batched forward passes (icl_set_file_options),
// so NetMutex is kept only for field compatibility and is not taken.
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
