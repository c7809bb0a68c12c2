// Package iclengine owns the ONE libimageclust_hip.so context the drop-in packages share (internal/embeddings holds the
// model in it, internal/clustering clusters on it): a process that loads both packages creates one GPU context and one
// set of workspaces, not two.  NOT COMPILED in the authoring container (no Go toolchain): logic-free glue.
package iclengine

/*
#cgo CFLAGS: -I${SRCDIR}/../../../include
#cgo LDFLAGS: -L${SRCDIR}/../../../imageclust_amd -limageclust_hip
#include <stdlib.h>
#include "imageclust.h"
*/
import "C"

import (
	"fmt"
	"os"
	"runtime"
	"strconv"
	"sync"
	"unsafe"
)

var (
	once sync.Once
	ctx  *C.icl_ctx
	err  error
)

// Ctx returns the shared context as an unsafe.Pointer (cgo types are package-local: each caller casts it back to its
// own *C.icl_ctx), creating it on first use.  ICL_DEVICE in the environment selects the GPU (icl_create's device ordinal 0 otherwise).
func Ctx() (unsafe.Pointer, error) {
	once.Do(func() {
		// icl_last_error(NULL) reads a thread-local string: keep the failing call and the read on one OS thread
		runtime.LockOSThread()
		defer runtime.UnlockOSThread()
		dev := 0
		if v, e := strconv.Atoi(os.Getenv("ICL_DEVICE")); e == nil && v >= 0 {
			dev = v
		}
		if rc := C.icl_create(C.int(dev), &ctx); rc != C.ICL_OK {
			err = fmt.Errorf("icl_create: %s", C.GoString(C.icl_last_error(nil)))
		}
	})
	return unsafe.Pointer(ctx), err
}

var (
	modelMu     sync.Mutex
	modelLoaded string // path of the ONNX file whose weights the shared context holds ("" = none)
)

// LoadModelOnce loads the ONNX file into the shared context unless that very file is already loaded: the reference calls
// LoadPretrainedModelONNX once per request (workflow.go), and reloading would stall every embedder in flight behind the context's mutex.
func LoadModelOnce(path string) error {
	raw, e := Ctx()
	if e != nil {
		return e
	}
	modelMu.Lock()
	defer modelMu.Unlock()
	if modelLoaded == path {
		return nil
	}
	p := C.CString(path)
	defer C.free(unsafe.Pointer(p))
	if rc := C.icl_model_load_onnx((*C.icl_ctx)(raw), p); rc != C.ICL_OK {
		return fmt.Errorf("%s", C.GoString(C.icl_last_error((*C.icl_ctx)(raw))))
	}
	modelLoaded = path
	return nil
}

// LastError returns the message of the last failed call on the context (stored in the context, not in TLS).
func LastError() string {
	if ctx == nil {
		return "no context"
	}
	return C.GoString(C.icl_last_error(ctx))
}
