// Package clustering: drop-in replacement of imageclust/internal/clustering backed by libimageclust_hip.so.
// Same exported names and signatures as the reference's clustering.go; the arithmetic runs on the GPU through
// the C-ABI of include/imageclust.h.  NOT COMPILED in the authoring container (no Go toolchain): logic-free glue.
package clustering

/*
#cgo CFLAGS: -I${SRCDIR}/../../../include
#cgo LDFLAGS: -L${SRCDIR}/../../../imageclust_amd -limageclust_hip
#include <stdlib.h>
#include "imageclust.h"
*/
import "C"

import (
	"fmt"
	"log"
	"sync"
	"unsafe"
)

// Cluster mirrors clustering.go:11-15.
type Cluster struct {
	Indices  []int
	Size     int
	Centroid []float32
}

var (
	ctxOnce sync.Once
	ctx     *C.icl_ctx
	ctxErr  error
)

func engine() (*C.icl_ctx, error) {
	ctxOnce.Do(func() {
		if rc := C.icl_create(0, &ctx); rc != C.ICL_OK {
			ctxErr = fmt.Errorf("icl_create: %s", C.GoString(C.icl_last_error(nil)))
		}
	})
	return ctx, ctxErr
}

// NewCluster: clustering.go:18-26 (plain Go, unchanged).
func NewCluster(index int, embedding []float32) Cluster {
	c := make([]float32, len(embedding))
	copy(c, embedding)
	return Cluster{Indices: []int{index}, Size: 1, Centroid: c}
}

// RemoveClusters: clustering.go:51-58 (plain Go, unchanged).
func RemoveClusters(clusters []Cluster, i, j int) []Cluster {
	if i > j {
		i, j = j, i
	}
	clusters = append(clusters[:j], clusters[j+1:]...)
	clusters = append(clusters[:i], clusters[i+1:]...)
	return clusters
}

// MergeClusters: clustering.go:29-47; the centroid comes from icl_merge_centroid.
func MergeClusters(a, b Cluster) Cluster {
	e, err := engine()
	if err != nil {
		log.Panic(err)
	}
	out := make([]float32, len(a.Centroid))
	if len(out) > 0 {
		C.icl_merge_centroid(e, (*C.float)(unsafe.Pointer(&a.Centroid[0])), C.int64_t(a.Size),
			(*C.float)(unsafe.Pointer(&b.Centroid[0])), C.int64_t(b.Size), C.int32_t(len(out)), (*C.float)(unsafe.Pointer(&out[0])))
	}
	return Cluster{Indices: append(append([]int{}, a.Indices...), b.Indices...), Size: a.Size + b.Size, Centroid: out}
}

func flatten(clusters []Cluster) (flat []float32, sizes []int32, d int) {
	if len(clusters) == 0 {
		return nil, nil, 0
	}
	d = len(clusters[0].Centroid)
	flat = make([]float32, len(clusters)*d) // [][]float32 cannot cross cgo: one contiguous buffer
	sizes = make([]int32, len(clusters))
	for i, c := range clusters {
		copy(flat[i*d:(i+1)*d], c.Centroid)
		sizes[i] = int32(c.Size)
	}
	return
}

// ComputeInitialDistanceMatrix: clustering.go:61-73 -> icl_ward_distance_matrix.
func ComputeInitialDistanceMatrix(clusters []Cluster) [][]float32 {
	n := len(clusters)
	out := make([][]float32, n)
	if n == 0 {
		return out
	}
	e, err := engine()
	if err != nil {
		log.Panic(err)
	}
	flat, sizes, d := flatten(clusters)
	D := make([]float32, n*n)
	var cptr *C.float
	if d > 0 {
		cptr = (*C.float)(unsafe.Pointer(&flat[0]))
	}
	if rc := C.icl_ward_distance_matrix(e, cptr, (*C.int32_t)(unsafe.Pointer(&sizes[0])), C.int64_t(n), C.int32_t(d),
		(*C.float)(unsafe.Pointer(&D[0])), C.int64_t(n)); rc != C.ICL_OK {
		log.Panicf("icl_ward_distance_matrix: %s", C.GoString(C.icl_last_error(e)))
	}
	for i := range out {
		out[i] = D[i*n : (i+1)*n : (i+1)*n]
	}
	return out
}

// FindClosestClusters: clustering.go:119-133 -> icl_find_closest.
func FindClosestClusters(distanceMatrix [][]float32) (int, int) {
	n := len(distanceMatrix)
	if n < 2 {
		return -1, -1
	}
	e, err := engine()
	if err != nil {
		log.Panic(err)
	}
	flat := make([]float32, n*n)
	for i, row := range distanceMatrix {
		copy(flat[i*n:(i+1)*n], row)
	}
	var i, j C.int64_t
	if rc := C.icl_find_closest(e, (*C.float)(unsafe.Pointer(&flat[0])), C.int64_t(n), C.int64_t(n), &i, &j); rc != C.ICL_OK {
		log.Panicf("icl_find_closest: %s", C.GoString(C.icl_last_error(e)))
	}
	return int(i), int(j)
}

// WardDistance: clustering.go:136-145 (a 2x2 call of the exact distance tile).
func WardDistance(a, b Cluster) float32 {
	return ComputeInitialDistanceMatrix([]Cluster{a, b})[1][0]
}

// CalculateOptimalClusters: clustering.go:168-186.
func CalculateOptimalClusters(totalItems, minSize, maxSize int) (int, error) {
	var k C.int64_t
	if rc := C.icl_calc_optimal_clusters(C.int64_t(totalItems), C.int64_t(minSize), C.int64_t(maxSize), &k); rc != C.ICL_OK {
		if totalItems < minSize {
			return 0, fmt.Errorf("total items (%d) less than minimum cluster size (%d)", totalItems, minSize)
		}
		return 0, fmt.Errorf("cannot satisfy cluster size constraints with total items (%d), minSize (%d), and maxSize (%d)", totalItems, minSize, maxSize)
	}
	return int(k), nil
}

// PerformClusteringWithConstraints: clustering.go:198-284 -> icl_cluster (exact update: bit-identical ids).
func PerformClusteringWithConstraints(embeddings [][]float32, productReferenceIDs []string, minSize, maxSize int) (map[int][]string, bool) {
	n := len(embeddings)
	log.Printf("Total items for clustering: %d", n)
	e, err := engine()
	if err != nil {
		log.Printf("Clustering engine error: %v", err)
		return nil, false
	}
	d := 0
	if n > 0 {
		d = len(embeddings[0])
	}
	flat := make([]float32, n*d)
	for i, row := range embeddings {
		copy(flat[i*d:(i+1)*d], row)
	}
	cid := make([]int32, n+1)
	rank := make([]int32, n+1)
	var nc C.int32_t
	var eptr *C.float
	if len(flat) > 0 {
		eptr = (*C.float)(unsafe.Pointer(&flat[0]))
	}
	rc := C.icl_cluster(e, eptr, C.int64_t(n), C.int32_t(d), C.int32_t(minSize), C.int32_t(maxSize), C.ICL_UPDATE_EXACT,
		(*C.int32_t)(unsafe.Pointer(&cid[0])), (*C.int32_t)(unsafe.Pointer(&rank[0])), &nc)
	if rc != C.ICL_OK {
		log.Printf("Clustering constraint error: %s", C.GoString(C.icl_last_error(e)))
		return nil, false
	}
	clusterMap := make(map[int][]string, int(nc))
	sizes := make([]int, int(nc))
	for i := 0; i < n; i++ {
		if cid[i] >= 0 {
			sizes[cid[i]]++
		}
	}
	for c, s := range sizes {
		clusterMap[c] = make([]string, s)
	}
	for i := 0; i < n; i++ {
		if cid[i] >= 0 {
			clusterMap[int(cid[i])][rank[i]] = productReferenceIDs[i]
		}
	}
	log.Printf("Clustering successful. Formed %d valid clusters.", len(clusterMap))
	return clusterMap, true
}
