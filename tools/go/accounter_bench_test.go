// accounter_bench_test.go — the reference's own CPU path timed on the bench workload, for anyone with a Go
// toolchain (the build image has none, so this file is NOT compiled or run here: bench.py's cpu_baseline is
// the C restatement in oracle/, kind "port").
//
// Drop it into pkg/flow/ of netobserv-ebpf-agent and run on the GPU node's host:
//
//	go test ./pkg/flow/ -run xxx -bench BenchmarkAccounterZipf -benchtime 1x
//
// It feeds Accounter.Account through its channels exactly as pkg/flow/account_test.go:47-60 does
// (one producer goroutine = the ring-buffer reader, one Accounter goroutine), with the population of
// pkg/model/bench_fixtures_test.go:19-50 and Zipf(1.1) ranks over 1 M flows (BASELINE configs[1]).
package flow

import (
	"math/rand"
	"net"
	"testing"
	"time"

	"github.com/netobserv/netobserv-ebpf-agent/pkg/ebpf"
	"github.com/netobserv/netobserv-ebpf-agent/pkg/metrics"
	"github.com/netobserv/netobserv-ebpf-agent/pkg/model"
)

const (
	zipfFlows   = 1_000_000
	zipfRecords = 20_000_000 // the sample bench.py times the restatement on
)

func zipfFlowID(i int) ebpf.BpfFlowId {
	var id ebpf.BpfFlowId
	copy(id.SrcIp[:], net.IPv4(10, byte(i>>16), byte(i>>8), byte(i)).To16())
	copy(id.DstIp[:], net.IPv4(10, byte(i>>16), byte(i>>8), byte(i+1)).To16())
	id.SrcPort = uint16(1024 + (i % 60000))
	id.DstPort = 443
	id.TransportProtocol = 6
	return id
}

func BenchmarkAccounterZipf(b *testing.B) {
	rng := rand.New(rand.NewSource(2))
	zipf := rand.NewZipf(rng, 1.1, 1, zipfFlows-1)
	ids := make([]ebpf.BpfFlowId, zipfFlows)
	for i := range ids {
		ids[i] = zipfFlowID(i)
	}
	records := make([]model.RawRecord, zipfRecords)
	for j := range records {
		i := int(zipf.Uint64())
		records[j] = model.RawRecord{Id: ids[i], Metrics: ebpf.BpfFlowMetrics{
			StartMonoTimeTs: uint64(1_000_000 + j), EndMonoTimeTs: uint64(2_000_000 + j),
			Bytes: uint64(1500 * (1 + j%10)), Packets: uint32(1 + j%10), EthProtocol: 0x0800, Flags: 0x10,
			SrcMac: [6]uint8{2, 0, 0, 0, 0, 1}, DstMac: [6]uint8{2, 0, 0, 0, 0, 2}, IfIndexFirstSeen: uint32(2 + i%4),
		}}
	}
	now := time.Now()
	b.ResetTimer()
	for n := 0; n < b.N; n++ {
		acc := NewAccounter(1<<27, time.Hour, func() time.Time { return now },
			func() time.Duration { return 3_000_000 }, metrics.NoOp(), nil, false)
		inputs := make(chan *model.RawRecord, 50) // BUFFERS_LENGTH default
		evictor := make(chan []*model.Record, 1)
		go acc.Account(inputs, evictor)
		for j := range records {
			inputs <- &records[j]
		}
		close(inputs) // the closing eviction (account.go:73-80)
		flows := <-evictor
		b.ReportMetric(float64(len(records))/b.Elapsed().Seconds()/1e6*float64(n+1), "Mrecords/s")
		_ = flows
	}
}
