graph [
  directed 1
  node [
    id 0
    label "0"
  ]
  node [
    id 1
    label "1"
  ]
  node [
    id 2
    label "2"
  ]
  node [
    id 3
    label "3"
  ]
  node [
    id 4
    label "4"
  ]
  node [
    id 5
    label "5"
  ]
  node [
    id 6
    label "6"
  ]
  node [
    id 7
    label "7"
  ]
  node [
    id 8
    label "8"
  ]
  node [
    id 9
    label "9"
  ]
  node [
    id 10
    label "10"
  ]
  node [
    id 11
    label "11"
  ]
  node [
    id 12
    label "12"
  ]
  node [
    id 13
    label "13"
  ]
  edge [
    source 0
    target 1
    capacity 4429
  ]
  edge [
    source 0
    target 2
    capacity 2925
  ]
  edge [
    source 0
    target 7
    capacity 4873
  ]
  edge [
    source 1
    target 2
    capacity 1052
  ]
  edge [
    source 1
    target 3
    capacity 1435
  ]
  edge [
    source 1
    target 0
    capacity 4429
  ]
  edge [
    source 2
    target 5
    capacity 4247
  ]
  edge [
    source 2
    target 0
    capacity 2925
  ]
  edge [
    source 2
    target 1
    capacity 1052
  ]
  edge [
    source 3
    target 4
    capacity 4383
  ]
  edge [
    source 3
    target 10
    capacity 4933
  ]
  edge [
    source 3
    target 1
    capacity 1435
  ]
  edge [
    source 4
    target 5
    capacity 3425
  ]
  edge [
    source 4
    target 6
    capacity 1716
  ]
  edge [
    source 4
    target 3
    capacity 4383
  ]
  edge [
    source 5
    target 9
    capacity 4818
  ]
  edge [
    source 5
    target 12
    capacity 4974
  ]
  edge [
    source 5
    target 2
    capacity 4247
  ]
  edge [
    source 5
    target 4
    capacity 3425
  ]
  edge [
    source 6
    target 7
    capacity 2901
  ]
  edge [
    source 6
    target 4
    capacity 1716
  ]
  edge [
    source 7
    target 8
    capacity 2779
  ]
  edge [
    source 7
    target 0
    capacity 4873
  ]
  edge [
    source 7
    target 6
    capacity 2901
  ]
  edge [
    source 8
    target 9
    capacity 2699
  ]
  edge [
    source 8
    target 11
    capacity 1458
  ]
  edge [
    source 8
    target 13
    capacity 4185
  ]
  edge [
    source 8
    target 7
    capacity 2779
  ]
  edge [
    source 9
    target 5
    capacity 4818
  ]
  edge [
    source 9
    target 8
    capacity 2699
  ]
  edge [
    source 10
    target 11
    capacity 2985
  ]
  edge [
    source 10
    target 13
    capacity 1731
  ]
  edge [
    source 10
    target 3
    capacity 4933
  ]
  edge [
    source 11
    target 12
    capacity 3251
  ]
  edge [
    source 11
    target 8
    capacity 1458
  ]
  edge [
    source 11
    target 10
    capacity 2985
  ]
  edge [
    source 12
    target 13
    capacity 3414
  ]
  edge [
    source 12
    target 5
    capacity 4974
  ]
  edge [
    source 12
    target 11
    capacity 3251
  ]
  edge [
    source 13
    target 8
    capacity 4185
  ]
  edge [
    source 13
    target 10
    capacity 1731
  ]
  edge [
    source 13
    target 12
    capacity 3414
  ]
]
