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
    capacity 1067
  ]
  edge [
    source 0
    target 2
    capacity 1778
  ]
  edge [
    source 0
    target 7
    capacity 1548
  ]
  edge [
    source 1
    target 2
    capacity 1068
  ]
  edge [
    source 1
    target 3
    capacity 4812
  ]
  edge [
    source 1
    target 0
    capacity 1067
  ]
  edge [
    source 2
    target 5
    capacity 4551
  ]
  edge [
    source 2
    target 0
    capacity 1778
  ]
  edge [
    source 2
    target 1
    capacity 1068
  ]
  edge [
    source 3
    target 4
    capacity 1340
  ]
  edge [
    source 3
    target 10
    capacity 4116
  ]
  edge [
    source 3
    target 1
    capacity 4812
  ]
  edge [
    source 4
    target 5
    capacity 4212
  ]
  edge [
    source 4
    target 6
    capacity 2241
  ]
  edge [
    source 4
    target 3
    capacity 1340
  ]
  edge [
    source 5
    target 9
    capacity 4194
  ]
  edge [
    source 5
    target 12
    capacity 4027
  ]
  edge [
    source 5
    target 2
    capacity 4551
  ]
  edge [
    source 5
    target 4
    capacity 4212
  ]
  edge [
    source 6
    target 7
    capacity 1036
  ]
  edge [
    source 6
    target 4
    capacity 2241
  ]
  edge [
    source 7
    target 8
    capacity 3987
  ]
  edge [
    source 7
    target 0
    capacity 1548
  ]
  edge [
    source 7
    target 6
    capacity 1036
  ]
  edge [
    source 8
    target 9
    capacity 2333
  ]
  edge [
    source 8
    target 11
    capacity 3792
  ]
  edge [
    source 8
    target 13
    capacity 2180
  ]
  edge [
    source 8
    target 7
    capacity 3987
  ]
  edge [
    source 9
    target 5
    capacity 4194
  ]
  edge [
    source 9
    target 8
    capacity 2333
  ]
  edge [
    source 10
    target 11
    capacity 4275
  ]
  edge [
    source 10
    target 13
    capacity 1130
  ]
  edge [
    source 10
    target 3
    capacity 4116
  ]
  edge [
    source 11
    target 12
    capacity 2568
  ]
  edge [
    source 11
    target 8
    capacity 3792
  ]
  edge [
    source 11
    target 10
    capacity 4275
  ]
  edge [
    source 12
    target 13
    capacity 3598
  ]
  edge [
    source 12
    target 5
    capacity 4027
  ]
  edge [
    source 12
    target 11
    capacity 2568
  ]
  edge [
    source 13
    target 8
    capacity 2180
  ]
  edge [
    source 13
    target 10
    capacity 1130
  ]
  edge [
    source 13
    target 12
    capacity 3598
  ]
]
