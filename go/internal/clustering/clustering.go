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
	"unsafe"

	"imageclust/internal/iclengine"
)

// Cluster mirrors clustering.go:11-15.
type Cluster struct {
	Indices  []int
	Size     int
	Centroid []float32
}

// engine: the context shared with internal/embeddings (one GPU context and workspace per process).
func engine() (*C.icl_ctx, error) {
	p, err := iclengine.Ctx()
	return (*C.icl_ctx)(p), err
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
		if rc := C.icl_merge_centroid(e, (*C.float)(unsafe.Pointer(&a.Centroid[0])), C.int64_t(a.Size),
			(*C.float)(unsafe.Pointer(&b.Centroid[0])), C.int64_t(b.Size), C.int32_t(len(out)), (*C.float)(unsafe.Pointer(&out[0]))); rc != C.ICL_OK {
			log.Panicf("icl_merge_centroid: %s", C.GoString(C.icl_last_error(e))) // never a silent zero centroid
		}
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

// UpdateDistanceMatrix: clustering.go:76-96 -> icl_update_distance_matrix (rows/columns removedIdx1, removedIdx2 dropped
// order-preserving, new last row/column from the centroids).  clusters is the list AFTER RemoveClusters + append (:240-241).
func UpdateDistanceMatrix(distanceMatrix [][]float32, clusters []Cluster, newCluster Cluster, removedIdx1, removedIdx2 int) [][]float32 {
	n := len(distanceMatrix)
	e, err := engine()
	if err != nil {
		log.Panic(err)
	}
	flatD := make([]float32, n*n)
	for i, row := range distanceMatrix {
		copy(flatD[i*n:(i+1)*n], row)
	}
	flatC, sizes, d := flatten(clusters)
	m := n - 1
	out := make([]float32, m*m)
	var cptr *C.float
	if d > 0 {
		cptr = (*C.float)(unsafe.Pointer(&flatC[0]))
	}
	if rc := C.icl_update_distance_matrix(e, (*C.float)(unsafe.Pointer(&flatD[0])), C.int64_t(n), C.int64_t(n), cptr,
		(*C.int32_t)(unsafe.Pointer(&sizes[0])), C.int32_t(d), C.int64_t(removedIdx1), C.int64_t(removedIdx2),
		(*C.float)(unsafe.Pointer(&out[0])), C.int64_t(m)); rc != C.ICL_OK {
		log.Panicf("icl_update_distance_matrix: %s", C.GoString(C.icl_last_error(e)))
	}
	res := make([][]float32, m)
	for i := range res {
		res[i] = out[i*m : (i+1)*m : (i+1)*m]
	}
	return res
}

// RemoveRowsAndColumns: clustering.go:100-116 (plain Go, unchanged).
func RemoveRowsAndColumns(matrix [][]float32, i, j int) [][]float32 {
	if i > j {
		i, j = j, i
	}
	for idx := range matrix {
		matrix[idx] = append(matrix[idx][:j], matrix[idx][j+1:]...)
		matrix[idx] = append(matrix[idx][:i], matrix[idx][i+1:]...)
	}
	matrix = append(matrix[:j], matrix[j+1:]...)
	matrix = append(matrix[:i], matrix[i+1:]...)
	return matrix
}

// DotFloat32: clustering.go:148-157 (plain Go, unchanged: in-order fp32 sum, panics on length mismatch).
func DotFloat32(a, b []float32) float32 {
	if len(a) != len(b) {
		panic("Vectors must be the same length")
	}
	var sum float32
	for i := range a {
		sum += a[i] * b[i]
	}
	return sum
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
